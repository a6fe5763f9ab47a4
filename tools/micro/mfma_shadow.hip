// How many non-MFMA instructions does one wave per SIMD hide under back-to-back v_mfma_f32_32x32x2_f32 (64 cycles each)?
// 16 independent accumulators (the Winograd kernel's register shape), NV independent VALU adds and NL ds_read_b128 pinned behind
// every MFMA (the reads fire-and-forget, waited for once per 16 MFMAs).  Answers whether the Winograd K loop (about 4.5 other instructions per MFMA) is
// issue-limited in principle.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_shadow tools/micro/mfma_shadow.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, int KIND = 0>
__global__ __launch_bounds__(256) void shadow(float* out, int iters, float a0, float b0) {
    extern __shared__ float lds[];      // 100 KiB requested at launch: one workgroup per CU = one wave per SIMD
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 * i + threadIdx.x;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 xp[16];
    for (int i = 0; i < 16; ++i) xp[i] = f32x2{a0 * i, b0 + threadIdx.x};
    const f32x2 bp = {b0, a0};
    unsigned sx[8];
    for (int i = 0; i < 8; ++i) sx[i] = __builtin_amdgcn_readfirstlane((unsigned)iters * i);
    f32x4 ld[4] = {};
    const float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = a0 * i;
    __syncthreads();
    const unsigned laddr = threadIdx.x * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < (KIND >= 3 ? 0 : NV); ++v) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(k * NV + v) & 15]) : "v"(b0));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[(k * NV + v) & 15]) : "v"(bp));
                if (KIND == 2) asm volatile("s_add_u32 %0, %0, 7" : "+s"(sx[(k * NV + v) & 7]));
                if (KIND == 7) asm volatile("s_nop 0");
                if (KIND == 8) asm volatile("s_nop 3");
            }
            // KIND 3 / 4: the same VALU count in bursts — 4 * NV adds behind every fourth MFMA (4: plus one s_add behind the others)
            if (KIND >= 3) {
                if ((k & 3) == 3) {
#pragma unroll
                    for (int v = 0; v < 4 * NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[v & 15]) : "v"(b0));
                } else if (KIND == 4) {
                    asm volatile("s_add_u32 %0, %0, 7" : "+s"(sx[k & 7]));
                }
            }
            // KIND 5 / 6: bursts of 8 * NV behind every eighth MFMA (6: s_add behind the others)
            if (KIND >= 5) {
            }
#pragma unroll
            for (int l = 0; l < NL; ++l)      // fire and forget (waited for once per 16 MFMAs): the issue cost and the LDS bandwidth, not the latency
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[(k * NL + l) & 3]) : "v"(laddr), "n"((((k * NL + l) & 15) * 4096) & 0xffff));
            __builtin_amdgcn_sched_barrier(0);      // hard pin: this MFMA's companions stay behind it
        }
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += x[i] + xp[i].x + xp[i].y;
    for (int i = 0; i < 8; ++i) s += (float)sx[i];
    for (int i = 0; i < 4; ++i) s += ld[i][0] + ld[i][1] + ld[i][2] + ld[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NL, int KIND = 0>
void run() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount, iters = 2000;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&shadow<NV, NL, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((shadow<NV, NL, KIND>), dim3(blocks), dim3(256), 100 * 1024, 0, out, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((shadow<NV, NL, KIND>), dim3(blocks), dim3(256), 100 * 1024, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 16.0 * 4096.0;
    printf("per MFMA: %2d %s + %d ds_read_b128 : %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", NV, KIND == 0 ? "v_add_f32" : (KIND == 1 ? "v_pk_add_f32" : (KIND == 2 ? "s_add_u32" : KIND == 7 ? "s_nop 0" : KIND == 8 ? "s_nop 3" : (KIND == 3 ? "v_add_f32 (in bursts behind every 4th MFMA)" : "v_add_f32 (bursts behind every 4th MFMA, s_add behind the others)"))), NL, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 157.3);
    hipFree(out);
}

int main() {
    run<0, 0>();
    run<2, 0>();
    run<4, 0>();
    run<6, 0>();
    run<8, 0>();
    run<12, 0>();
    run<16, 0>();
    run<0, 1>();
    run<0, 2>();
    run<4, 1>();
    run<2, 0, 1>();
    run<4, 0, 1>();
    run<8, 0, 1>();
    run<4, 0, 2>();
    run<8, 0, 2>();
    run<1, 0, 2>();
    run<1, 0, 7>();
    run<2, 0, 7>();
    run<1, 0, 8>();
    run<2, 0, 3>();
    run<4, 0, 3>();
    run<8, 0, 3>();
    run<2, 0, 4>();
    run<4, 0, 4>();
    run<8, 0, 4>();
    return 0;
}
