// Operand / result lane layout of v_mfma_f32_16x16x4_f32 (checked before building a kernel on it):
//   A[m][k]: lane l holds m = l % 16, k = l / 16;  B[k][n]: k = l / 16, n = l % 16;  D[i][j]: lane l, register r: i = 4 * (l / 16) + r, j = l % 16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = acc[r];
}
int main() {
    float hA[64], hB[64], hD[256], *dA, *dB, *dD;
    for (int i = 0; i < 64; ++i) hA[i] = (float)(rand() % 17 - 8), hB[i] = (float)(rand() % 13 - 6);
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float w = 0;
            for (int kk = 0; kk < 4; ++kk) w += hA[i * 4 + kk] * hB[kk * 16 + j];
            bad += w != hD[i * 16 + j];
        }
    printf("mfma_f32_16x16x4 layout check: %d mismatches of 256\n", bad);
    return bad != 0;
}
