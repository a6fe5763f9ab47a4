// Sustained rate of v_mfma_f32_16x16x4_f32 with 32 independent f32x4 accumulators (the two-wave Winograd kernel's shape), at one and
// two waves per SIMD, accumulators left to the compiler (arch VGPRs) — is the instruction itself the limit of conv_wino16_kernel?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int WAVES, int FORM>
__global__ __launch_bounds__(64 * WAVES * 4) void rate(float* out, int iters, float a0, float b0) {
    extern __shared__ float lds[];
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) a[i] = a0 + threadIdx.x * 1e-3f + i, b[i] = b0 - threadIdx.x * 1e-3f - i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                if (FORM == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + h) & 7], b[(i * 3 + h) & 7], acc[i], 0, 0, 0);
                if (FORM == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[(i + h) & 7]), "v"(b[(i * 3 + h) & 7]));
                if (FORM == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + h) & 7]), "v"(b[(i * 3 + h) & 7]));
            }
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x & 1];
}
template <int WAVES, int FORM>
void run() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount, iters = 2000, threads = 64 * WAVES * 4;
    float* out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&rate<WAVES, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<WAVES, FORM>), dim3(blocks), dim3(threads), 100 * 1024, 0, out, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<WAVES, FORM>), dim3(blocks), dim3(threads), 100 * 1024, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * WAVES * iters * 64.0 * 2048.0;
    printf("v_mfma_f32_16x16x4_f32, %d wave(s) per SIMD, 32 accumulators, %s: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", WAVES, FORM == 0 ? "builtin (compiler's choice)" : FORM == 1 ? "accumulators in AGPRs" : "accumulators in arch VGPRs", ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    hipFree(out);
}
int main() {
    run<1, 0>();
    run<2, 0>();
    run<2, 1>();
    run<2, 2>();
    run<1, 2>();
    return 0;
}
