// Microbenchmark: what does back-to-back v_mfma_f32_32x32x2_f32 sustain on this chip?  No memory traffic at all.
// NACC independent accumulators per wave, WAVES waves per SIMD (via block count / launch bounds).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// operands that differ per lane and per MFMA (pseudo-random bit patterns, |x| ~ 1): the toggling of real activations
template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop_rand(float* out, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av[16], bv[16];
    unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int k = 0; k < 16; ++k) {
        h = h * 1664525u + 1013904223u;
        av[k] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;      // U(-0.25, 0.25)
        h = h * 1664525u + 1013904223u;
        bv[k] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(k + i) & 15], bv[(k + 3 * i) & 15], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run_rand(int blocks_per_cu) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * blocks_per_cu, iters = 4000;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop_rand<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1u);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop_rand<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 16.0 * NACC * 4096.0;
        printf("RANDOM operands NACC=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", NACC, blocks_per_cu, ms, flop / ms / 1e9,
               flop / ms / 1e9 / 157.3);
    }
    hipFree(out);
}

template <int NACC>
void run(int blocks_per_cu, float scale) {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * blocks_per_cu, iters = 4000;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f * scale, 0.5f * scale);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f * scale, 0.5f * scale);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 /*waves*/ * iters * 16.0 * NACC * 4096.0;
    printf("NACC=%d waves/SIMD=%d data_scale=%g: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", NACC, blocks_per_cu, scale, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 157.3);
    hipFree(out);
}

int main() {
    run<4>(1, 1.0f);
    run<4>(2, 1.0f);
    run<2>(2, 1.0f);
    run<4>(2, 0.0f);   // all-zero operands: data-dependent power
    run<8>(1, 1.0f);
    run_rand<4>(2);
    run_rand<4>(1);
    run_rand<8>(1);
    return 0;
}
