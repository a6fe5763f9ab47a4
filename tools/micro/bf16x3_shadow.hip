// What would the Winograd K loop cost if its 64 v_mfma_f32_32x32x2_f32 per 8-channel chunk (4096 matrix-pipe cycles, sharing the
// SIMD's fp32 datapath with every VALU instruction of the wave) were 48 v_mfma_f32_32x32x16_bf16 on a 3-way bf16 split of both
// operands (a1b1 + a2b1 | a1b2 + a2b2 | a3b1 + a1b3: two products per K = 16 MFMA, fp32-level accuracy), with the split's VALU work
// (~5 instructions per transformed value) and the B-fragment reads beside them?  One wave per SIMD, 16 accumulators of 32x32 as in
// conv_wino_kernel; per "chunk" 48 MFMAs, NV VALU instructions and NL ds_read_b64 behind every MFMA.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/bf16x3_shadow tools/micro/bf16x3_shadow.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int NV, int NL, int MIX>
__global__ __launch_bounds__(256) void shadow(float* out, int iters, float a0, float b0) {
    extern __shared__ float lds[];
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x2 xp[16];
    for (int i = 0; i < 16; ++i) xp[i] = f32x2{a0 * i, b0 + threadIdx.x};
    unsigned xi[16];
    for (int i = 0; i < 16; ++i) xi[i] = threadIdx.x * 77u + i;
    const f32x2 bp = {b0, a0};
    bf16x8 A[3], B[3];
    for (int t = 0; t < 3; ++t)
        for (int k = 0; k < 8; ++k) A[t][k] = (__bf16)(a0 * (t + 1) + k + threadIdx.x * 1e-3f), B[t][k] = (__bf16)(b0 - t - k * 0.1f);
    f32x2 ld[4] = {};
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = a0 * i;
    __syncthreads();
    const unsigned laddr = threadIdx.x * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[t], B[t], acc[k], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int j = (k * NV + v) & 15;
                    if (MIX == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[j]) : "v"(bp));
                    if (MIX == 1) {      // the split's mix: cvt_pk, shift, and, pk_add (2 : 2 : 2 : 4)
                        const int m = (k * NV + v) % 5;
                        if (m == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(xi[j]) : "v"(xp[j].x), "v"(xp[j].y));
                        if (m == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(xi[(j + 1) & 15]) : "v"(xi[j]));
                        if (m == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(xi[(j + 2) & 15]) : "v"(xi[j]));
                        if (m >= 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[j]) : "v"(bp));
                    }
                }
#pragma unroll
                for (int l = 0; l < NL; ++l)
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ld[(k * NL + l) & 3]) : "v"(laddr), "n"((((k * NL + l) & 15) * 2048) & 0xffff));
                __builtin_amdgcn_sched_barrier(0);
            }
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += xp[i].x + xp[i].y + (float)xi[i];
    for (int i = 0; i < 4; ++i) s += ld[i][0] + ld[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV, int NL, int MIX>
void run() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount, iters = 2000;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&shadow<NV, NL, MIX>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((shadow<NV, NL, MIX>), dim3(blocks), dim3(256), 100 * 1024, 0, out, 10, 1.0f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((shadow<NV, NL, MIX>), dim3(blocks), dim3(256), 100 * 1024, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // one iteration = one 8-channel "chunk" of the Winograd K loop (48 MFMAs).  The fp32 kernel's chunk: 64 MFMAs x 64 cycles = 4096 cycles of matrix pipe
    const double us_per_chunk = ms * 1e3 / iters;
    printf("per MFMA: %2d VALU (%s) + %d ds_read_b64 : %.3f ms  %.3f us per 48-MFMA chunk = %.0f cycles at 2.4 GHz (fp32 form: 4096 cycles of matrix pipe alone, ~6100 measured)\n",
           NV, MIX ? "split mix" : "v_pk_add_f32", NL, ms, us_per_chunk, us_per_chunk * 2400.0);
    hipFree(out);
}

int main() {
    run<0, 0, 0>();
    run<2, 0, 0>();
    run<4, 0, 0>();
    run<6, 0, 0>();
    run<8, 0, 0>();
    run<10, 0, 0>();
    run<4, 2, 1>();
    run<6, 2, 1>();
    run<8, 2, 1>();
    run<9, 2, 1>();
    run<10, 2, 1>();
    run<8, 0, 1>();
    return 0;
}
