// M2M's two custom ops as HIP kernels for gfx950 (NHWC fp32).
//
//   softsplat (summation forward warp)  replaces softsplat_out, vfi_models/ops/cupy_ops/softsplat.py:140-192
//   9x9 mean-L1 cost volume             replaces costvol_out,   vfi_models/ops/cupy_ops/costvol.py:4-43
//
// Differences from the CUDA originals that do not change the arithmetic:
//   * splat: owner-computes tiles with LDS accumulation instead of one global atomic per contribution
//     (see the comment at the kernel); flow read and bilinear weights once per source pixel, not per channel;
//   * cost volume: the 9x9 neighbourhood of `two` comes from an LDS tile with a 4-pixel halo (zero outside the
//     image: |one - 0| is exactly the reference's out-of-bounds branch), channels accumulated in the
//     reference's order, so the result is bit-identical to the sequential restatement.
#include "vfi_common.h"

#include <cstdlib>

namespace vfi {

// ---- summation splat -----------------------------------------------------------------------------------
// Device-scope fp32 atomics on MI355X are resolved beyond the per-XCD L2s (8 non-coherent L2s), and a splat is
// nothing but atomics: the straightforward kernel ran at 54 GB/s (profiles/r01_splat_bench_v1.txt).  So each
// workgroup OWNS a 32x32 output tile instead: it scans the source window that can reach the tile
// (tile dilated by R = ceil(max|flow|)+1, capped at SPLAT_RCAP), accumulates in LDS (ds_add_f32) and writes
// the tile once with plain coalesced stores.  Sources displaced by more than the cap are rare; a second pass
// adds them with global atomics.  max|flow| is reduced on the device — no host synchronisation.
// Measured (profiles/r01_splat_bench_v2.txt, [1,1088,1920,4]): 210-295 us per launch vs 1230-1530 us for the
// one-global-atomic-per-contribution form; of that ~135 us is ds_add_f32 itself (a racy plain-RMW experiment
// ran in 49 us) and the rest the window scan, which grows with max|flow|.  Workgroup-scope GLOBAL fp32 atomics
// into the owned tile were also tried: no faster than device scope (1.0-1.6 ms).
constexpr int SPLAT_T = 32;
constexpr int SPLAT_RCAP = 64;
constexpr int SPLAT_CMAX = 8;   // channels per pass (LDS tile 32x32x8 floats = 32 KiB)

__global__ void flow_absmax_kernel(const float* __restrict__ flow, long n2, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        const float v = fabsf(flow[i]);
        if (isfinite(v)) m = fmaxf(m, v);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)  // one device-scope atomic per workgroup; non-negative floats order as uints
        atomicMax(out_bits, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

__device__ static inline void splat_weights(float fx, float fy, int& x0, int& y0, float (&w)[4]) {
    x0 = (int)floorf(fx);
    y0 = (int)floorf(fy);
    const float x1 = (float)(x0 + 1), y1 = (float)(y0 + 1);
    w[0] = __fmul_rn(__fsub_rn(x1, fx), __fsub_rn(y1, fy));                 // north-west
    w[1] = __fmul_rn(__fsub_rn(fx, (float)x0), __fsub_rn(y1, fy));          // north-east
    w[2] = __fmul_rn(__fsub_rn(x1, fx), __fsub_rn(fy, (float)y0));          // south-west
    w[3] = __fmul_rn(__fsub_rn(fx, (float)x0), __fsub_rn(fy, (float)y0));   // south-east
}

// LDS accumulator layout is planar, acc[c][pixel] (64 consecutive targets = 64 different banks per ds_add_f32); NCT > 0
// fixes the channel count at compile time (M2M: 4, one float4 load per source).  Measured against the interleaved
// [pixel][c] layout: no change (179 vs 180 us per [1,1088,1920,4] splat, profiles/r01b_splat_bench_v3.txt) — the kernel is
// bound by the ds_add_f32 rate itself: 37.8 M atomic lanes in 179 us = 0.34 lanes per clock per CU, independent of the
// flow magnitude (sigma 0 ... 8 px), of bank conflicts and of the window size.
template <int NCT>
__global__ __launch_bounds__(256) void softsplat_tile_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                             float* __restrict__ out, const unsigned* __restrict__ absmax_bits,
                                                             int H, int W, int C, int c0, int nc_rt, int tiles_x, int tiles_y) {
    constexpr int PLANE = SPLAT_T * SPLAT_T;
    const int nc = NCT > 0 ? NCT : nc_rt;
    __shared__ float acc[PLANE * (NCT > 0 ? NCT : SPLAT_CMAX)];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x - n * tiles_x * tiles_y;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int X0 = tx * SPLAT_T, Y0 = ty * SPLAT_T;
    for (int i = tid; i < SPLAT_T * SPLAT_T * nc; i += 256) acc[i] = 0.f;
    __syncthreads();
    const float amax = __uint_as_float(*absmax_bits);
    int R = (int)ceilf(amax) + 1;
    if (R > SPLAT_RCAP) R = SPLAT_RCAP;
    const float cap = (float)(SPLAT_RCAP - 1);
    const int wx0 = max(X0 - R, 0), wx1 = min(X0 + SPLAT_T + R, W);
    const int wy0 = max(Y0 - R, 0), wy1 = min(Y0 + SPLAT_T + R, H);
    const int ww = wx1 - wx0, wh = wy1 - wy0;
    const size_t nbase = (size_t)n * H * W;
    // window scan, 4 source pixels per thread and iteration so that 4 flow loads are in flight together
    constexpr int U = 4;
    for (int i0 = tid; i0 < ww * wh; i0 += 256 * U) {
        float2 fl[U];
        int sxs[U], sys[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
            ok[u] = i < ww * wh;
            const int ii = ok[u] ? i : 0;
            sys[u] = wy0 + ii / ww;
            sxs[u] = wx0 + ii % ww;
            fl[u] = ((const float2*)flow)[nbase + (size_t)sys[u] * W + sxs[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sx = sxs[u], sy = sys[u];
            const float2 f = fl[u];
            const size_t sp = nbase + (size_t)sy * W + sx;
            const float fx = (float)sx + f.x, fy = (float)sy + f.y;
            // softsplat.py:157-158 (non-finite targets are skipped); far pixels go to the second pass
            bool live = ok[u] && isfinite(fx) && isfinite(fy) && !(fmaxf(fabsf(f.x), fabsf(f.y)) > cap);
            int x0 = 0, y0 = 0;
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            if (live) splat_weights(fx, fy, x0, y0, w);
            const int lx = x0 - X0, ly = y0 - Y0;                         // tile-local north-west target
            live = live && !(lx < -1 || lx >= SPLAT_T || ly < -1 || ly >= SPLAT_T);
            const float* ip = in + sp * C + c0;
            if (NCT == 4) {
                // ---- 4 channels, with neighbour hand-off.  The 64 lanes scan 64 consecutive source pixels; where the
                // flow is smooth, lane l's eastern targets are lane l+1's western targets.  Then lane l hands its NE / SE
                // contributions to lane l+1 (DPP row shift, 16-lane rows), which adds them to its own NW / SW before the
                // LDS atomic: 8 instead of 16 ds_add_f32 per source — the instruction this kernel is bound by.
                float4 iv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) iv = (C & 3) == 0 && (c0 & 3) == 0 ? *(const float4*)ip : make_float4(ip[0], ip[1], ip[2], ip[3]);
                const int key = live ? ((ly + 1) << 8) + (lx + 1) : -4096;           // target cell id (row, col)
                const int keyL = __builtin_amdgcn_update_dpp(-8192, key, 0x111, 0xf, 0xf, false);   // from lane-1 (row_shr:1)
                const int keyR = __builtin_amdgcn_update_dpp(-8192, key, 0x101, 0xf, 0xf, false);   // from lane+1 (row_shl:1)
                const bool take = live && keyL + 1 == key;       // left neighbour's east column == my west column
                const bool give = live && key + 1 == keyR;       // the right neighbour takes my east column
                float e[8] = {iv.x * w[1], iv.y * w[1], iv.z * w[1], iv.w * w[1], iv.x * w[3], iv.y * w[3], iv.z * w[3], iv.w * w[3]};
                float fromL[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    fromL[j] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e[j]), 0x111, 0xf, 0xf, false));
                float wv[8] = {iv.x * w[0], iv.y * w[0], iv.z * w[0], iv.w * w[0], iv.x * w[2], iv.y * w[2], iv.z * w[2], iv.w * w[2]};
                if (take) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) wv[j] += fromL[j];
                }
                if (live) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if ((k & 1) && give) continue;            // east column handed to the right neighbour
                        const int tx_ = lx + (k & 1), ty_ = ly + (k >> 1);
                        if (tx_ < 0 || tx_ >= SPLAT_T || ty_ < 0 || ty_ >= SPLAT_T || X0 + tx_ >= W || Y0 + ty_ >= H) continue;
                        float* a = &acc[ty_ * SPLAT_T + tx_];
                        const float* v = (k & 1) ? &e[(k >> 1) * 4] : &wv[(k >> 1) * 4];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            __hip_atomic_fetch_add(a + c * PLANE, v[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            } else if (live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int tx_ = lx + (k & 1), ty_ = ly + (k >> 1);
                    // inside this tile and inside the image (the reference's per-target bounds check)
                    if (tx_ < 0 || tx_ >= SPLAT_T || ty_ < 0 || ty_ >= SPLAT_T || X0 + tx_ >= W || Y0 + ty_ >= H) continue;
                    float* a = &acc[ty_ * SPLAT_T + tx_];
                    for (int c = 0; c < nc; ++c)
                        __hip_atomic_fetch_add(a + c * PLANE, __fmul_rn(ip[c], w[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < SPLAT_T * SPLAT_T * nc; i += 256) {
        const int pix = i / nc, c = i - pix * nc;
        const int y = Y0 + pix / SPLAT_T, x = X0 + pix % SPLAT_T;
        if (y < H && x < W) out[(nbase + (size_t)y * W + x) * C + c0 + c] = acc[c * PLANE + pix];
    }
}

// second pass: sources displaced by more than the cap (rare) -> device-scope atomics, as the CUDA original
__global__ void softsplat_far_kernel(const float* __restrict__ in, const float* __restrict__ flow, float* __restrict__ out,
                                     const unsigned* __restrict__ absmax_bits, int N, int H, int W, int C) {
    if (__uint_as_float(*absmax_bits) <= (float)(SPLAT_RCAP - 1)) return;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * H * W) return;
    const int x = idx % W, y = (idx / W) % H;
    const long nbase = idx - ((long)y * W + x);
    const float2 f = ((const float2*)flow)[idx];
    const float fx = (float)x + f.x, fy = (float)y + f.y;
    if (!isfinite(fx) || !isfinite(fy)) return;
    if (!(fmaxf(fabsf(f.x), fabsf(f.y)) > (float)(SPLAT_RCAP - 1))) return;
    int x0, y0;
    float w[4];
    splat_weights(fx, fy, x0, y0, w);
    const float* ip = in + idx * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tx = x0 + (k & 1), ty = y0 + (k >> 1);
        if (tx < 0 || tx >= W || ty < 0 || ty >= H) continue;
        float* o = out + (nbase + (long)ty * W + tx) * C;
        for (int c = 0; c < C; ++c) unsafeAtomicAdd(o + c, __fmul_rn(ip[c], w[k]));
    }
}

int softsplat_sum_launch(const float* in, const float* flow, float* out, int N, int H, int W, int C, hipStream_t s) {
    static unsigned* d_absmax = nullptr;  // one in-flight call per process
    if (!d_absmax) VFI_CHECK_HIP(hipMalloc((void**)&d_absmax, sizeof(unsigned)));
    const long px = (long)N * H * W;
    VFI_CHECK_HIP(hipMemsetAsync(d_absmax, 0, sizeof(unsigned), s));
    {
        TraceScope ts("splat_absmax", s);
        hipLaunchKernelGGL(flow_absmax_kernel, dim3(256), dim3(256), 0, s, flow, px * 2, d_absmax);
    }
    const int tiles_x = cdiv(W, SPLAT_T), tiles_y = cdiv(H, SPLAT_T);
    for (int c0 = 0; c0 < C; c0 += SPLAT_CMAX) {
        const int nc = C - c0 < SPLAT_CMAX ? C - c0 : SPLAT_CMAX;
        TraceScope ts("softsplat_sum", s);
        if (nc == 4)
            hipLaunchKernelGGL(softsplat_tile_kernel<4>, dim3(N * tiles_x * tiles_y), dim3(256), 0, s, in, flow, out, d_absmax, H,
                               W, C, c0, nc, tiles_x, tiles_y);
        else
            hipLaunchKernelGGL(softsplat_tile_kernel<0>, dim3(N * tiles_x * tiles_y), dim3(256), 0, s, in, flow, out, d_absmax, H,
                               W, C, c0, nc, tiles_x, tiles_y);
    }
    {
        TraceScope ts("splat_far", s);
        hipLaunchKernelGGL(softsplat_far_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, s, in, flow, out, d_absmax,
                           N, H, W, C);
    }
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- cost volume -----------------------------------------------------------------------------------------
constexpr int CV_T = 16;            // output tile 16x16 pixels = 256 threads
constexpr int CV_TW = CV_T + 8;     // + 4-pixel halo each side
constexpr int CV_CK = 16;           // channels per LDS pass
constexpr int CV_S = CV_CK + 4;     // LDS pixel stride (floats)

__global__ __launch_bounds__(256) void costvol_kernel(const float* __restrict__ one, int one_cs,
                                                      const float* __restrict__ two, int two_cs, int two_swap,
                                                      float* __restrict__ out, int N, int H, int W, int C, int out_cs,
                                                      int out_coff, int tiles_x, int tiles_y) {
    __shared__ __attribute__((aligned(16))) float lds[CV_TW * CV_TW * CV_S];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x - n * tiles_x * tiles_y;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int X0 = tx * CV_T, Y0 = ty * CV_T;
    const int lx = tid & 15, ly = tid >> 4;
    const int x = X0 + lx, y = Y0 + ly;
    const bool inb = x < W && y < H;
    const int n2 = two_swap ? (n ^ 1) : n;  // `two` taken from the partner image of a (0,1) batch pair
    float acc[81];
#pragma unroll
    for (int k = 0; k < 81; ++k) acc[k] = 0.f;
    for (int c0 = 0; c0 < C; c0 += CV_CK) {
        __syncthreads();
        // stage `two` tile (zero outside the image)
        for (int i = tid; i < CV_TW * CV_TW * (CV_CK / 4); i += 256) {
            const int pix = i / (CV_CK / 4), q = i - pix * (CV_CK / 4);
            const int py = pix / CV_TW, px = pix - py * CV_TW;
            const int iy = Y0 - 4 + py, ix = X0 - 4 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H && ix >= 0 && ix < W && c0 + q * 4 < C)
                v = *(const float4*)(two + ((size_t)(n2 * H + iy) * W + ix) * two_cs + c0 + q * 4);
            *(float4*)&lds[pix * CV_S + q * 4] = v;
        }
        float o[CV_CK];
#pragma unroll
        for (int q = 0; q < CV_CK / 4; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (inb && c0 + q * 4 < C) v = *(const float4*)(one + ((size_t)(n * H + y) * W + x) * one_cs + c0 + q * 4);
            o[4 * q] = v.x;
            o[4 * q + 1] = v.y;
            o[4 * q + 2] = v.z;
            o[4 * q + 3] = v.w;
        }
        __syncthreads();
        const int nc = C - c0 < CV_CK ? C - c0 : CV_CK;
#pragma unroll
        for (int dy = 0; dy < 9; ++dy)
#pragma unroll
            for (int dx = 0; dx < 9; ++dx) {
                const float* t = &lds[((ly + dy) * CV_TW + lx + dx) * CV_S];
                float a = acc[dy * 9 + dx];
#pragma unroll
                for (int q = 0; q < CV_CK / 4; ++q) {
                    const float4 v = *(const float4*)(t + q * 4);
                    if (4 * q + 0 < nc) a = __fadd_rn(a, fabsf(__fsub_rn(o[4 * q + 0], v.x)));
                    if (4 * q + 1 < nc) a = __fadd_rn(a, fabsf(__fsub_rn(o[4 * q + 1], v.y)));
                    if (4 * q + 2 < nc) a = __fadd_rn(a, fabsf(__fsub_rn(o[4 * q + 2], v.z)));
                    if (4 * q + 3 < nc) a = __fadd_rn(a, fabsf(__fsub_rn(o[4 * q + 3], v.w)));
                }
                acc[dy * 9 + dx] = a;
            }
    }
    if (inb) {
        float* op = out + ((size_t)(n * H + y) * W + x) * out_cs + out_coff;
        const float fc = (float)C;
#pragma unroll
        for (int k = 0; k < 81; ++k) op[k] = __fdiv_rn(acc[k], fc);
    }
}

int costvol_launch(const float* one, int one_cs, const float* two, int two_cs, int two_swap, float* out, int N, int H, int W,
                   int C, int out_cs, int out_coff, hipStream_t s) {
    const int tiles_x = cdiv(W, CV_T), tiles_y = cdiv(H, CV_T);
    TraceScope ts("costvol9x9", s);
    hipLaunchKernelGGL(costvol_kernel, dim3(N * tiles_x * tiles_y), dim3(256), 0, s, one, one_cs, two, two_cs, two_swap, out, N,
                       H, W, C, out_cs, out_coff, tiles_x, tiles_y);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vfi

using namespace vfi;
#include "../../include/vfi_hip.h"

extern "C" {

int vfi_softsplat_sum(const float* in_dev, const float* flow_dev, float* out_dev, int N, int H, int W, int C, void* stream) {
    VFI_REQUIRE(in_dev && flow_dev && out_dev && N > 0 && H > 0 && W > 0 && C > 0, "vfi_softsplat_sum: bad arguments");
    VFI_REQUIRE(in_dev != out_dev, "vfi_softsplat_sum: in-place not supported");
    return softsplat_sum_launch(in_dev, flow_dev, out_dev, N, H, W, C, (hipStream_t)stream);
}

int vfi_costvol9x9(const float* one_dev, int one_cs, const float* two_dev, int two_cs, int two_swap, float* out_dev, int N,
                   int H, int W, int C, int out_cs, int out_coff, void* stream) {
    VFI_REQUIRE(one_dev && two_dev && out_dev && N > 0 && H > 0 && W > 0, "vfi_costvol9x9: bad arguments");
    VFI_REQUIRE(one_cs >= C && two_cs >= C && one_cs % 4 == 0 && two_cs % 4 == 0 && (!two_swap || N % 2 == 0),
                "vfi_costvol9x9: bad strides (one_cs=%d two_cs=%d) or odd batch with two_swap", one_cs, two_cs);
    VFI_REQUIRE(C > 0 && C % 4 == 0, "vfi_costvol9x9: C=%d must be a multiple of 4", C);
    VFI_REQUIRE(out_cs >= out_coff + 81 && out_coff >= 0, "vfi_costvol9x9: out_cs=%d cannot hold 81 channels at offset %d",
                out_cs, out_coff);
    return costvol_launch(one_dev, one_cs, two_dev, two_cs, two_swap, out_dev, N, H, W, C, out_cs, out_coff, (hipStream_t)stream);
}

}  // extern "C"
