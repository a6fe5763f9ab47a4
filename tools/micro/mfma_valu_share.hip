// Do fp32 MFMAs and VALU instructions of DIFFERENT waves on one SIMD run side by side, or do they share the SIMD's fp32 datapath?
// 512-thread workgroups, one per CU: two waves per SIMD (roles by HW_REG_HW_ID.SIMD_ID + an LDS ticket: the first wave of a SIMD is
// the matrix wave, the second the vector wave).  Timed: matrix waves alone, vector waves alone, both — independent pipes give
// T(both) ~ max, a shared one T(both) ~ sum.  Matrix: v_mfma_f32_32x32x2_f32 (fp32) or v_mfma_f32_32x32x16_bf16; vector: v_fma_f32,
// v_pk_fma_f32 or v_add_u32 (integer).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_valu_share tools/micro/mfma_valu_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MK, int VK>      // MK 0: fp32 MFMA, 1: bf16 MFMA;  VK 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_add_u32
__global__ __launch_bounds__(512) void share(float* out, int iters, int run_m, int run_v, float a0, float b0) {
    __shared__ int tick[4];
    if (threadIdx.x < 4) tick[threadIdx.x] = 0;
    __syncthreads();
    const int simd = (int)(__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4)) & 3;      // HW_REG_HW_ID[5:4]
    int t0 = 0;
    if ((threadIdx.x & 63) == 0) t0 = atomicAdd(&tick[simd], 1);
    const int role = __builtin_amdgcn_readfirstlane(t0) & 1;                                  // 0 matrix, 1 vector
    float s = 0.f;
    if (role == 0) {
        if (!run_m) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        const float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
        bf16x8 A, B;
        for (int k = 0; k < 8; ++k) A[k] = (__bf16)(a + k), B[k] = (__bf16)(b - k);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (MK == 0) acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 3], 0, 0, 0);
                else acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[k & 3], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        if (!run_v) return;
        float x[16];
        f32x2 xp[16];
        unsigned xi[16];
        for (int i = 0; i < 16; ++i) x[i] = a0 * i + threadIdx.x, xp[i] = f32x2{a0 * i, b0 + threadIdx.x}, xi[i] = threadIdx.x + i;
        const f32x2 bp = {b0, a0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                if (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k & 15]) : "v"(a0), "v"(b0));
                if (VK == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xp[k & 15]) : "v"(bp));
                if (VK == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(xi[k & 15]) : "v"(iters));
            }
        }
        for (int i = 0; i < 16; ++i) s += x[i] + xp[i].x + xp[i].y + (float)xi[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MK, int VK>
void run(const char* mname, const char* vname) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount, iters = 4000;
    float* out;
    (void)hipMalloc(&out, (size_t)blocks * 512 * 4);
    float ms[3];
    for (int c = 0; c < 3; ++c) {
        const int rm = c != 1, rv = c != 0;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((share<MK, VK>), dim3(blocks), dim3(512), 0, 0, out, 10, rm, rv, 1.0f, 0.5f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((share<MK, VK>), dim3(blocks), dim3(512), 0, 0, out, iters, rm, rv, 1.0f, 0.5f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[c], e0, e1);
    }
    // per SIMD and iteration: 16 MFMAs (64 cycles each fp32 / 32 bf16) and 64 VALU instructions
    printf("%-28s + %-13s: matrix alone %.3f ms, vector alone %.3f ms, both %.3f ms  -> both / max = %.2f, both / sum = %.2f\n", mname, vname, ms[0], ms[1], ms[2],
           ms[2] / (ms[0] > ms[1] ? ms[0] : ms[1]), ms[2] / (ms[0] + ms[1]));
    (void)hipFree(out);
}

int main() {
    run<0, 0>("v_mfma_f32_32x32x2_f32", "v_fma_f32");
    run<0, 1>("v_mfma_f32_32x32x2_f32", "v_pk_fma_f32");
    run<0, 2>("v_mfma_f32_32x32x2_f32", "v_add_u32");
    run<1, 0>("v_mfma_f32_32x32x16_bf16", "v_fma_f32");
    run<1, 1>("v_mfma_f32_32x32x16_bf16", "v_pk_fma_f32");
    run<1, 2>("v_mfma_f32_32x32x16_bf16", "v_add_u32");
    return 0;
}
