// A stand-in for a collective's kernel: `grid` workgroups of 256 threads that stay resident for `cycles` shader cycles.
#include <hip/hip_runtime.h>
__global__ void hog_kernel(long long cycles, int* sink) {
    const long long t0 = wall_clock64();
    int x = 0;
    while (wall_clock64() - t0 < cycles) x += 1;
    if (x == -1) *sink = x;
}
extern "C" int launch_hog(int grid, long long cycles, void* stream, int* sink) {
    hipLaunchKernelGGL(hog_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, cycles, sink);
    return (int)hipGetLastError();
}
