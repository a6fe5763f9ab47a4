// Known-traffic kernels for calibrating the rocprofv3 HBM counters on gfx950 (FETCH_SIZE / WRITE_SIZE units and the x2 rule of
// MI355X_MICROARCH.md's HBM section): each kernel moves an exactly known number of bytes over buffers far larger than the 256 MB
// Infinity Cache, so the counter value per launch can be compared with the truth.
//   calib_read   : reads  N bytes (float4 loads, grid-stride), writes 4 bytes per workgroup
//   calib_write  : writes N bytes, reads nothing
//   calib_copy   : reads  N and writes N
//   calib_read2x : reads the same N bytes twice within one launch (second pass after the first: 1 GiB apart = no cache reuse)
//   calib_read_b32 : reads N bytes with 4-byte loads per lane (256 B per wave instruction instead of 1 KiB)
//   calib_read_lds : reads N bytes with LDS-DMA (buffer_load_dwordx4 ... lds, the Winograd / direct conv kernels' operand path)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/hbm_known_traffic tools/micro/hbm_known_traffic.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_read(const f4* __restrict__ src, float* __restrict__ out, size_t n4, int passes) {
    float s = 0.f;
    for (int p = 0; p < passes; ++p)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            const f4 v = __builtin_nontemporal_load(src + i);
            s += v.x + v.y + v.z + v.w;
        }
    if (s == 12345.678f) out[blockIdx.x] = s;   // never true for the fill pattern: the loads stay, nothing is written
}
__global__ __launch_bounds__(256) void calib_read_b32(const float* __restrict__ src, float* __restrict__ out, size_t n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += __builtin_nontemporal_load(src + i);
    if (s == 12345.678f) out[blockIdx.x] = s;
}
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ __launch_bounds__(256) void calib_read_lds(const float* __restrict__ src, float* __restrict__ out, size_t bytes) {
    __shared__ float buf[4][4][256];                        // per wave: 4 pieces of 1 KiB in flight
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t per_block = (bytes / gridDim.x) & ~(size_t)16383;      // whole 16 KiB rounds (4 waves x 4 pieces x 1 KiB)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)src + blockIdx.x * per_block), 0, (int)per_block, 0x00020000);
    float s = 0.f;
    for (size_t off = 0; off < per_block; off += 16384) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)&buf[wave][i][0], 16, lane * 16, (int)off + (wave * 4 + i) * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s += buf[wave][lane & 3][lane];
    }
    if (s == 12345.678f) out[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void calib_write(f4* __restrict__ dst, size_t n4, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(f4{v, v, v, v}, dst + i);
}
__global__ __launch_bounds__(256) void calib_copy(const f4* __restrict__ src, f4* __restrict__ dst, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 1024) << 20;   // MiB
    const size_t n4 = bytes / 16;
    f4 *a, *b;
    float* o;
    CK(hipMalloc((void**)&a, bytes));
    CK(hipMalloc((void**)&b, bytes));
    CK(hipMalloc((void**)&o, 1 << 20));
    CK(hipMemset(a, 0x3c, bytes));
    CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 16;
    auto timed = [&](const char* name, double moved, auto&& launch) {
        launch();   // warm
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-14s true bytes per launch %.0f  %.3f ms  %.2f TB/s\n", name, moved, ms / 5, moved / (ms / 5) / 1e9);
    };
    timed("calib_read", (double)bytes, [&] { hipLaunchKernelGGL(calib_read, dim3(grid), dim3(256), 0, 0, a, o, n4, 1); });
    timed("calib_read2x", 2.0 * bytes, [&] { hipLaunchKernelGGL(calib_read, dim3(grid), dim3(256), 0, 0, a, o, n4, 2); });
    timed("calib_read_b32", (double)bytes, [&] { hipLaunchKernelGGL(calib_read_b32, dim3(grid), dim3(256), 0, 0, (const float*)a, o, bytes / 4); });
    {
        const size_t per_block = (bytes / grid) & ~(size_t)16383;
        timed("calib_read_lds", (double)per_block * grid, [&] { hipLaunchKernelGGL(calib_read_lds, dim3(grid), dim3(256), 0, 0, (const float*)a, o, bytes); });
    }
    timed("calib_write", (double)bytes, [&] { hipLaunchKernelGGL(calib_write, dim3(grid), dim3(256), 0, 0, b, n4, 1.5f); });
    timed("calib_copy", 2.0 * bytes, [&] { hipLaunchKernelGGL(calib_copy, dim3(grid), dim3(256), 0, 0, a, b, n4); });
    CK(hipDeviceSynchronize());
    return 0;
}
