// Launch side of the "body" kernels (gmfss_bodies.h, ifunet_bodies.h): one thread per output element,
//     __global__ k(Args a) { body(a, blockIdx.x * blockDim.x + threadIdx.x); }
// Built with -DVFI_HOSTCHECK (tests/hostcheck) run<>() loops over the same body on the host instead of launching, which is
// how the CPU test suite checks every body without a GPU.
#pragma once
#include "vfi_common.h"

namespace {

template <class Args, void (*Body)(const Args&, long)>
__global__ __launch_bounds__(256) void body_kernel(Args a) {
    Body(a, (long)blockIdx.x * blockDim.x + threadIdx.x);
}

template <class Args, void (*Body)(const Args&, long)>
int run(const Args& a, long n, void* stream, const char* name) {
    if (n <= 0) return 0;
#ifdef VFI_HOSTCHECK
    (void)stream;
    (void)name;
    for (long i = 0; i < n; ++i) Body(a, i);
    return 0;
#else
    hipStream_t s = (hipStream_t)stream;
    vfi::TraceScope ts(name, s);
    hipLaunchKernelGGL((body_kernel<Args, Body>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
#endif
}

}  // namespace
