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
#include <map>
#include <mutex>

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

__device__ static inline void splat_weights(float fx, float fy, int& x0, int& y0, float (&w)[4]) {
    x0 = (int)floorf(fx);
    y0 = (int)floorf(fy);
    const float x1 = (float)(x0 + 1), y1 = (float)(y0 + 1);
    w[0] = __fmul_rn(__fsub_rn(x1, fx), __fsub_rn(y1, fy));                 // north-west
    w[1] = __fmul_rn(__fsub_rn(fx, (float)x0), __fsub_rn(y1, fy));          // north-east
    w[2] = __fmul_rn(__fsub_rn(x1, fx), __fsub_rn(fy, (float)y0));          // south-west
    w[3] = __fmul_rn(__fsub_rn(fx, (float)x0), __fsub_rn(fy, (float)y0));   // south-east
}

// Source window of a tile: the bounding box of the sources that can reach it, from the per-32x32-block flow RANGES of
// flow_blockrange_kernel ([fx_min, fx_max, fy_min, fy_max] over the block's near, finite sources) — a source s reaches the
// tile iff floor(s + f) lies in [X0-1, X0+31] x [Y0-1, Y0+31], so block b contributes its pixels inside
// (tile - [f_min, f_max]); blocks further than two away cannot reach it (|f| <= 63).  For a smooth field this is the tile
// shifted against the flow, about 34 x 34 whatever the flow magnitude; for an incoherent field the tile dilated by max|f|.
// Call with all 256 threads; rq = 4 x 32 ints of LDS.  The window is at most 160 x 160.
struct SplatWin {
    int x0, x1, y0, y1;   // [x0, x1) x [y0, y1), empty when x1 <= x0
};
__device__ static inline SplatWin tile_window(const float4* __restrict__ brange, int n, int ty, int tx, int tiles_x, int tiles_y,
                                              int H, int W, int* rq) {
    const int tid = threadIdx.x;
    if (tid < 32) {
        int wx0 = 1 << 30, wx1 = -(1 << 30), wy0 = 1 << 30, wy1 = -(1 << 30);
        if (tid < 25) {
            const int dy = tid / 5 - 2, dx = tid % 5 - 2;
            const int by = ty + dy, bx = tx + dx;
            if (by >= 0 && by < tiles_y && bx >= 0 && bx < tiles_x) {
                const float4 r = brange[(n * tiles_y + by) * tiles_x + bx];
                if (r.x <= r.y && r.z <= r.w) {
                    const int X0 = tx * 32, Y0 = ty * 32;
                    const int lx = max((int)floorf((float)(X0 - 1) - r.y) - 1, bx * 32);
                    const int hx = min((int)ceilf((float)(X0 + 32) - r.x) + 1, min(bx * 32 + 32, W));
                    const int ly = max((int)floorf((float)(Y0 - 1) - r.w) - 1, by * 32);
                    const int hy = min((int)ceilf((float)(Y0 + 32) - r.z) + 1, min(by * 32 + 32, H));
                    if (lx < hx && ly < hy) wx0 = lx, wx1 = hx, wy0 = ly, wy1 = hy;
                }
            }
        }
        rq[tid] = wx0, rq[32 + tid] = wx1, rq[64 + tid] = wy0, rq[96 + tid] = wy1;
    }
    __syncthreads();
    SplatWin w = {1 << 30, -(1 << 30), 1 << 30, -(1 << 30)};
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        w.x0 = min(w.x0, rq[i]);
        w.x1 = max(w.x1, rq[32 + i]);
        w.y0 = min(w.y0, rq[64 + i]);
        w.y1 = max(w.y1, rq[96 + i]);
    }
    if (w.x1 <= w.x0 || w.y1 <= w.y0) w = SplatWin{0, 0, 0, 0};
    return w;
}

// LDS accumulator layout is planar, acc[c][pixel] (64 consecutive targets = 64 different banks per ds_add_f32); NCT > 0
// fixes the channel count at compile time (M2M: 4, one float4 load per source).  Measured against the interleaved
// [pixel][c] layout: no change (179 vs 180 us per [1,1088,1920,4] splat, profiles/r01b_splat_bench_v3.txt) — the kernel is
// bound by the ds_add_f32 rate itself: 37.8 M atomic lanes in 179 us = 0.34 lanes per clock per CU, independent of the
// flow magnitude (sigma 0 ... 8 px), of bank conflicts and of the window size.
template <int NCT>
__global__ __launch_bounds__(256) void softsplat_tile_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                             float* __restrict__ out, const float4* __restrict__ brange,
                                                             const unsigned* __restrict__ run_if_above, unsigned threshold,
                                                             int H, int W, int C, int c0, int nc_rt, int tiles_x, int tiles_y) {
    if (run_if_above && !(*run_if_above > threshold)) return;     // fallback of the list kernel: only when its spill list overflowed
    constexpr int PLANE = SPLAT_T * SPLAT_T;
    const int nc = NCT > 0 ? NCT : nc_rt;
    __shared__ float acc[PLANE * (NCT > 0 ? NCT : SPLAT_CMAX)];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x - n * tiles_x * tiles_y;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int X0 = tx * SPLAT_T, Y0 = ty * SPLAT_T;
    for (int i = tid; i < SPLAT_T * SPLAT_T * nc; i += 256) acc[i] = 0.f;
    __shared__ int rq[128];
    const SplatWin win = tile_window(brange, n, ty, tx, tiles_x, tiles_y, H, W, rq);     // (its barrier also covers the zeroing above)
    const float cap = (float)(SPLAT_RCAP - 1);
    const int wx0 = win.x0, wy0 = win.y0;
    const int ww = win.x1 - win.x0, wh = win.y1 - win.y0;
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

// ---- the list splat: owner-computes GATHER, no floating-point atomics ---------------------------------------------------------
// The tile kernel above is bound by the ds_add_f32 issue rate (16 per source, 8 with the DPP hand-off; 0.34 lanes per clock per
// CU, profiles/r01b_splat_bench_v4.txt: 180-240 us per [1,1088,1920,4] splat = 5 % of the HBM roofline).  A splat needs no
// floating-point atomic at all if every OUTPUT pixel knows which sources reach it.  So each workgroup (still the owner of a
// 32x32 output tile) first files every source of its window under the tile cell its north-west target falls into — ONE
// integer ds_add_rtn_u32 per source and a 2-byte list entry (the source's window coordinates) — and then every output pixel
// gathers: it walks the lists of the four cells (p, p-1 in x and y) whose 2x2 footprints cover it, recomputes the reference's
// bilinear weights from the flow (cached loads) and accumulates in registers, in a fixed order (lists sorted by source raster
// position; south-east, south-west, north-east, north-west contribution — for a uniform translation exactly the ascending-source
// order of the sequential oracle, bit-identical; always bit-identical run to run).  One coalesced float4 store per pixel.  Any
// channel count walks the same lists.
//   * window: per-tile bounding box of the sources that can reach it, from a 32x32-block table of flow ranges (pre-pass);
//   * a cell holding more than SPLAT_K sources (strongly convergent flow) spills the excess to a global overflow list that a
//     small second kernel adds with global atomics; if that list overflows too (pathological fields), the whole launch is
//     redone by the LDS-atomic tile kernel above — always correct, never silently truncated;
//   * sources displaced by more than SPLAT_RCAP-1 px go to the far pass, as before.
constexpr int SPLAT_K = 8;
constexpr int SPLAT_GCAP = 4 * SPLAT_K;               // contributors of one output pixel: four cells x SPLAT_K (spills go to the tail)
constexpr int SPLAT_CW = SPLAT_T + 1;                 // cells per row: north-west targets lx in [-1, 31]
constexpr int SPLAT_CELLS = SPLAT_CW * SPLAT_CW;
constexpr unsigned SPLAT_OVF_CAP = 1u << 20;          // entries of the global overflow list (8 MiB); VFI_SPLAT_SPILL_CAP lowers it (tests)

struct SplatCtl {            // device-side control words, zeroed per call
    unsigned absmax_bits;    // max |flow| over the launch, only maintained above the cap (far pass switch)
    unsigned ovf_count;      // overflow entries appended (may exceed SPLAT_OVF_CAP: then the fallback runs)
};

// pre-pass: brange[n][ty][tx] = [fx_min, fx_max, fy_min, fy_max] over the block's near (|f|inf <= cap), finite sources
// (non-finite flows never splat; far ones go to the far pass); an empty block gets an inverted range
__global__ __launch_bounds__(256) void flow_blockrange_kernel(const float* __restrict__ flow, int H, int W, int tiles_x,
                                                              int tiles_y, float4* __restrict__ brange, SplatCtl* __restrict__ ctl) {
    const int tid = threadIdx.x;
    const int n = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x - n * tiles_x * tiles_y;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int x = tx * SPLAT_T + (tid & 31);
    const float big = 3.0e38f;
    float x0 = big, x1 = -big, y0 = big, y1 = -big, all = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int y = ty * SPLAT_T + (tid >> 5) + 8 * q;
        if (x < W && y < H) {
            const float2 f = ((const float2*)flow)[((size_t)n * H + y) * W + x];
            const float v = fmaxf(fabsf(f.x), fabsf(f.y));
            if (isfinite(f.x) && isfinite(f.y)) {
                all = fmaxf(all, v);
                if (!(v > (float)(SPLAT_RCAP - 1))) x0 = fminf(x0, f.x), x1 = fmaxf(x1, f.x), y0 = fminf(y0, f.y), y1 = fmaxf(y1, f.y);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o)), x1 = fmaxf(x1, __shfl_xor(x1, o));
        y0 = fminf(y0, __shfl_xor(y0, o)), y1 = fmaxf(y1, __shfl_xor(y1, o));
        all = fmaxf(all, __shfl_xor(all, o));
    }
    __shared__ float wr[4][5];
    if ((tid & 63) == 0) {
        float* r = wr[tid >> 6];
        r[0] = x0, r[1] = x1, r[2] = y0, r[3] = y1, r[4] = all;
    }
    __syncthreads();
    if (tid == 0) {
        brange[blockIdx.x] = make_float4(fminf(fminf(wr[0][0], wr[1][0]), fminf(wr[2][0], wr[3][0])),
                                         fmaxf(fmaxf(wr[0][1], wr[1][1]), fmaxf(wr[2][1], wr[3][1])),
                                         fminf(fminf(wr[0][2], wr[1][2]), fminf(wr[2][2], wr[3][2])),
                                         fmaxf(fmaxf(wr[0][3], wr[1][3]), fmaxf(wr[2][3], wr[3][3])));
        const float a = fmaxf(fmaxf(wr[0][4], wr[1][4]), fmaxf(wr[2][4], wr[3][4]));
        if (a > (float)(SPLAT_RCAP - 1)) atomicMax(&ctl->absmax_bits, __float_as_uint(a));   // rare: only far sources
    }
}

template <int NCT>   // 4: C == 4, float4 path (M2M);  0: any C, 4 channels per walk
__global__ __launch_bounds__(256) void softsplat_list_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                             float* __restrict__ out, const float4* __restrict__ brange,
                                                             SplatCtl* __restrict__ ctl, uint2* __restrict__ ovf, unsigned ovf_cap,
                                                             int H, int W, int C, int tiles_x, int tiles_y,
                                                             uint2* __restrict__ glist, unsigned char* __restrict__ gcount) {
#pragma clang fp contract(off)   // in * w, then +: the reference's atomicAdd(out, in * w) cannot fuse either
    __shared__ int cnt[SPLAT_CELLS];
    __shared__ unsigned short lst[SPLAT_CELLS * SPLAT_K];
    __shared__ int rq[128];
    const int tid = threadIdx.x;
    const int n = blockIdx.x / (tiles_x * tiles_y);
    const int trem = blockIdx.x - n * tiles_x * tiles_y;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int X0 = tx * SPLAT_T, Y0 = ty * SPLAT_T;
    for (int i = tid; i < SPLAT_CELLS; i += 256) cnt[i] = 0;
    const SplatWin win = tile_window(brange, n, ty, tx, tiles_x, tiles_y, H, W, rq);     // (its barrier also covers the zeroing above)
    const float cap = (float)(SPLAT_RCAP - 1);
    const int wx0 = win.x0, wy0 = win.y0;
    const int ww = win.x1 - win.x0, wh = win.y1 - win.y0;
    const size_t nbase = (size_t)n * H * W;
    // ---- phase 1: file every source of the window under its north-west target cell
    constexpr int U = 4;
    for (int i0 = tid; i0 < ww * wh; i0 += 256 * U) {
        float2 fl[U];
        int dxs[U], dys[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * 256;
            ok[u] = i < ww * wh;
            const int ii = ok[u] ? i : 0;
            dys[u] = ii / ww;
            dxs[u] = ii - dys[u] * ww;
            fl[u] = ((const float2*)flow)[nbase + (size_t)(wy0 + dys[u]) * W + wx0 + dxs[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sx = wx0 + dxs[u], sy = wy0 + dys[u];
            const float2 f = fl[u];
            const float fx = (float)sx + f.x, fy = (float)sy + f.y;
            // softsplat.py:157-158 (non-finite targets are skipped); far sources go to the far pass
            if (!(ok[u] && isfinite(fx) && isfinite(fy) && !(fmaxf(fabsf(f.x), fabsf(f.y)) > cap))) continue;
            const int lx = (int)floorf(fx) - X0, ly = (int)floorf(fy) - Y0;      // tile-local north-west target
            if (lx < -1 || lx >= SPLAT_T || ly < -1 || ly >= SPLAT_T) continue;
            const int cell = (ly + 1) * SPLAT_CW + lx + 1;
            const int slot = atomicAdd(&cnt[cell], 1);
            if (slot < SPLAT_K) {
                lst[cell * SPLAT_K + slot] = (unsigned short)((dys[u] << 8) | dxs[u]);     // window <= 160 x 160
            } else {
                const unsigned g = atomicAdd(&ctl->ovf_count, 1u);
                if (g < ovf_cap) ovf[g] = make_uint2((unsigned)(nbase + (size_t)sy * W + sx), blockIdx.x);
            }
        }
    }
    __syncthreads();
    // ---- phase 2: sort each list by source raster position (fixed summation order, run to run and vs the oracle)
    for (int c = tid; c < SPLAT_CELLS; c += 256) {
        const int k = min(cnt[c], SPLAT_K);
        unsigned short* l = &lst[c * SPLAT_K];
        for (int i = 1; i < k; ++i) {
            const unsigned short v = l[i];
            int j = i - 1;
            while (j >= 0 && l[j] > v) {
                l[j + 1] = l[j];
                --j;
            }
            l[j + 1] = v;
        }
    }
    __syncthreads();
    // ---- phase 3: gather
    if (NCT == 0) {
        // any channel count: this kernel only resolves, per output pixel, WHO contributes and with what weight — the (source,
        // weight) pairs in summation order, at most 4 cells x SPLAT_K of them, to glist[pixel][.] — and softsplat_gather_kernel
        // (one thread per output ELEMENT, coalesced over channels) does the arithmetic.  Walking the lists once per 4-channel
        // group from here, as round 2's first version did, recomputed the weights C/4 times per pixel and read `in` 16 bytes
        // per lane at a stride of 4C bytes: 0.9 ms per GMFSS feature splat, 3 % of its HBM bound.
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const int px = tid & 31, py = (tid >> 5) + 8 * q;
            const int x = X0 + px, y = Y0 + py;
            if (x >= W || y >= H) continue;
            const size_t pix = nbase + (size_t)y * W + x;
            uint2* gl = glist + pix * SPLAT_GCAP;
            int m = 0;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                const int cell = (py + 1 - (k >> 1)) * SPLAT_CW + px + 1 - (k & 1);
                const int ne = min(cnt[cell], SPLAT_K);
                for (int e = 0; e < ne; ++e) {
                    const unsigned ent = lst[cell * SPLAT_K + e];
                    const int sx = wx0 + (int)(ent & 255u), sy = wy0 + (int)(ent >> 8);
                    const size_t sp = nbase + (size_t)sy * W + sx;
                    const float2 f = ((const float2*)flow)[sp];
                    int x0, y0;
                    float w[4];
                    splat_weights((float)sx + f.x, (float)sy + f.y, x0, y0, w);
                    gl[m++] = make_uint2((unsigned)sp, __float_as_uint(w[k]));
                }
            }
            gcount[pix] = (unsigned char)m;
        }
        return;
    }
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int px = tid & 31, py = (tid >> 5) + 8 * q;
        const int x = X0 + px, y = Y0 + py;
        if (x >= W || y >= H) continue;
        float* op = out + (nbase + (size_t)y * W + x) * C;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 3; k >= 0; --k) {                           // k = dx + 2 dy: SE, SW, NE, NW contribution of the source
            const int cell = (py + 1 - (k >> 1)) * SPLAT_CW + px + 1 - (k & 1);
            const int ne = min(cnt[cell], SPLAT_K);
            for (int e = 0; e < ne; ++e) {
                const unsigned ent = lst[cell * SPLAT_K + e];
                const int sx = wx0 + (int)(ent & 255u), sy = wy0 + (int)(ent >> 8);
                const size_t sp = nbase + (size_t)sy * W + sx;
                const float2 f = ((const float2*)flow)[sp];
                int x0, y0;
                float w[4];
                splat_weights((float)sx + f.x, (float)sy + f.y, x0, y0, w);
                const float wk = w[k];
                const float4 iv = *(const float4*)(in + sp * C);
                a0 += iv.x * wk, a1 += iv.y * wk, a2 += iv.z * wk, a3 += iv.w * wk;
            }
        }
        *(float4*)op = make_float4(a0, a1, a2, a3);
    }
}

// the arithmetic of the any-channel-count splat: out[pixel][c] = sum over the pixel's list of in[source][c] * weight, in list
// order (= the order the one-kernel form summed in: results are bit-identical to it).  One thread per output element: stores
// are fully coalesced, loads are runs of C consecutive floats per source, the list entries are broadcast reads.
__global__ __launch_bounds__(256) void softsplat_gather_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               const uint2* __restrict__ glist,
                                                               const unsigned char* __restrict__ gcount, long total, int C) {
#pragma clang fp contract(off)
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const long pix = idx / C;
    const int c = (int)(idx - pix * C);
    const int m = gcount[pix];
    const uint2* gl = glist + pix * SPLAT_GCAP;
    float a = 0.f;
    for (int i = 0; i < m; ++i) {
        const uint2 e = gl[i];
        a += in[(size_t)e.x * C + c] * __uint_as_float(e.y);
    }
    out[idx] = a;
}

// the spilled sources of over-full cells: global atomics onto the tile the list kernel has already written
__device__ static inline void splat_spill_pass(const float* __restrict__ in, const float* __restrict__ flow, float* __restrict__ out,
                                               unsigned total, const uint2* __restrict__ ovf, int H, int W, int C, int tiles_x,
                                               int tiles_y) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint2 e = ovf[i];
        const size_t sp = e.x;
        const int n = e.y / (tiles_x * tiles_y);
        const int trem = e.y - n * tiles_x * tiles_y;
        const int X0 = (trem % tiles_x) * SPLAT_T, Y0 = (trem / tiles_x) * SPLAT_T;
        const size_t nbase = (size_t)n * H * W;
        const int sy = (int)((sp - nbase) / W), sx = (int)((sp - nbase) - (size_t)sy * W);
        const float2 f = ((const float2*)flow)[sp];
        int x0, y0;
        float w[4];
        splat_weights((float)sx + f.x, (float)sy + f.y, x0, y0, w);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tx = x0 + (k & 1), ty = y0 + (k >> 1);
            if (tx < X0 || tx >= X0 + SPLAT_T || ty < Y0 || ty >= Y0 + SPLAT_T || tx >= W || ty >= H) continue;   // this tile's share
            float* o = out + (nbase + (size_t)ty * W + tx) * C;
            for (int c = 0; c < C; ++c) unsafeAtomicAdd(o + c, __fmul_rn(in[sp * C + c], w[k]));
        }
    }
}

// sources displaced by more than the cap (rare) -> device-scope atomics, as the CUDA original
__device__ static inline void splat_far_pass(const float* __restrict__ in, const float* __restrict__ flow, float* __restrict__ out,
                                             int N, int H, int W, int C) {
    const long total = (long)N * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = idx % W, y = (idx / W) % H;
        const long nbase = idx - ((long)y * W + x);
        const float2 f = ((const float2*)flow)[idx];
        const float fx = (float)x + f.x, fy = (float)y + f.y;
        if (!isfinite(fx) || !isfinite(fy)) continue;
        if (!(fmaxf(fabsf(f.x), fabsf(f.y)) > (float)(SPLAT_RCAP - 1))) continue;
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
}

// one tail launch after the tile kernels: spilled sources (when the list kernel produced any and its list did not overflow —
// otherwise the fallback has redone the launch) and far sources; both usually absent, then every workgroup exits at once
__global__ __launch_bounds__(256) void softsplat_tail_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                                             float* __restrict__ out, const SplatCtl* __restrict__ ctl,
                                                             const uint2* __restrict__ ovf, unsigned ovf_cap, int do_spill, int N,
                                                             int H, int W, int C, int tiles_x, int tiles_y) {
    const unsigned spilled = ctl->ovf_count;
    if (do_spill && spilled > 0 && spilled <= ovf_cap) splat_spill_pass(in, flow, out, spilled, ovf, H, W, C, tiles_x, tiles_y);
    if (__uint_as_float(ctl->absmax_bits) > (float)(SPLAT_RCAP - 1)) splat_far_pass(in, flow, out, N, H, W, C);
}

// workspace of the splat (control words, block maxima, spill list) per (device, stream) — r6: engines of several pair lanes splat on
// the same device at the same time, each on its own stream; grows, never shrinks
struct SplatWs {
    SplatCtl* ctl = nullptr;
    uint2* ovf = nullptr;
    float4* brange = nullptr;
    size_t brange_n = 0;
    uint2* glist = nullptr;            // any-C path: [pixels][SPLAT_GCAP] (source, weight) pairs + their count per pixel
    unsigned char* gcount = nullptr;
    size_t g_pixels = 0;
};
static int splat_ws(hipStream_t s, size_t n_blocks, size_t list_pixels, SplatWs** out) {
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "splat: device index %d out of range", dev);
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, SplatWs> table;      // (node addresses are stable: the caller keeps the pointer)
    std::lock_guard<std::mutex> lock(mu);
    SplatWs& w = table[{dev, s}];
    if (!w.ctl) {
        VFI_CHECK_HIP(hipMalloc((void**)&w.ctl, sizeof(SplatCtl)));
        VFI_CHECK_HIP(hipMalloc((void**)&w.ovf, sizeof(uint2) * (size_t)SPLAT_OVF_CAP));
    }
    if (w.brange_n < n_blocks) {
        // outgrown blocks are RETIRED, not freed: a captured HIP graph of the stream's owner may have their addresses baked in (r6)
        VFI_CHECK_HIP(hipMalloc((void**)&w.brange, sizeof(float4) * n_blocks));
        w.brange_n = n_blocks;
    }
    if (w.g_pixels < list_pixels) {
        w.glist = nullptr, w.gcount = nullptr, w.g_pixels = 0;
        VFI_CHECK_HIP(hipMalloc((void**)&w.glist, sizeof(uint2) * SPLAT_GCAP * list_pixels));     // 256 B per pixel; only the used
        VFI_CHECK_HIP(hipMalloc((void**)&w.gcount, list_pixels));                                   // entries are ever touched
        w.g_pixels = list_pixels;
    }
    *out = &w;
    return 0;
}


int softsplat_sum_launch(const float* in, const float* flow, float* out, int N, int H, int W, int C, hipStream_t s) {
    // A/B options (tests of the fallback paths): splat_atomic = 1 forces the LDS-atomic tile kernel, splat_spill_cap shrinks the spill list
    // splat_atomic: 0 = default (the list kernel below), 1 = the LDS-atomic tile kernel, 3 = (C == 4) the staged list gather with source
    // compaction of m2m_render.hip — the M2M render kernel's machinery as a single splat: correct (tests/test_gpu_m2m_render.py) but a
    // tile's serial chain of load round trips makes it SLOWER than the list kernel as a stand-alone splat (profiles/r06_splat_bench.txt),
    // so it stays an A/B form
    const long mode_opt = option(kOptSplatAtomic);
    if (mode_opt == 3 && C == 4 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0 && ((uintptr_t)flow & 7) == 0) return softsplat4_launch(in, flow, out, N, H, W, s);
    const int g_splat_mode = mode_opt == 1 ? 1 : 0;
    const long cap_opt = option(kOptSplatSpillCap);
    const unsigned g_splat_cap = (cap_opt >= 0 && cap_opt < (long)SPLAT_OVF_CAP) ? (unsigned)cap_opt : SPLAT_OVF_CAP;
    const int tiles_x = cdiv(W, SPLAT_T), tiles_y = cdiv(H, SPLAT_T);
    const unsigned n_tiles = (unsigned)N * tiles_x * tiles_y;
    SplatWs* ws = nullptr;
    const bool c4 = C == 4 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    const size_t pixels = (size_t)N * H * W;
    VFI_REQUIRE(pixels < (1ull << 32), "softsplat: %zu pixels do not fit the 32-bit source index", pixels);
    if (int rc = splat_ws(s, n_tiles, g_splat_mode == 0 && !c4 ? pixels : 0, &ws)) return rc;
    VFI_CHECK_HIP(hipMemsetAsync(ws->ctl, 0, sizeof(SplatCtl), s));
    {
        TraceScope ts("splat_blockrange", s);
        hipLaunchKernelGGL(flow_blockrange_kernel, dim3(n_tiles), dim3(256), 0, s, flow, H, W, tiles_x, tiles_y, ws->brange, ws->ctl);
    }
    const unsigned* ovf_count = &ws->ctl->ovf_count;
    if (g_splat_mode == 0) {
        {
            TraceScope ts("softsplat_sum", s);
            if (c4)
                hipLaunchKernelGGL(softsplat_list_kernel<4>, dim3(n_tiles), dim3(256), 0, s, in, flow, out, ws->brange, ws->ctl, ws->ovf,
                                   g_splat_cap, H, W, C, tiles_x, tiles_y, (uint2*)nullptr, (unsigned char*)nullptr);
            else
                hipLaunchKernelGGL(softsplat_list_kernel<0>, dim3(n_tiles), dim3(256), 0, s, in, flow, out, ws->brange, ws->ctl, ws->ovf,
                                   g_splat_cap, H, W, C, tiles_x, tiles_y, ws->glist, ws->gcount);
        }
        if (!c4) {
            TraceScope ts("splat_gather", s);
            const long total = (long)pixels * C;
            hipLaunchKernelGGL(softsplat_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, ws->glist, ws->gcount,
                               total, C);
        }
    }
    // LDS-atomic tile kernel: the whole job when forced, otherwise only if the spill list overflowed (it then rewrites `out`)
    {
        TraceScope ts(g_splat_mode ? "softsplat_sum" : "splat_fallback", s);
        const unsigned* guard = g_splat_mode ? nullptr : ovf_count;
        for (int c0 = 0; c0 < C; c0 += SPLAT_CMAX) {
            const int nc = C - c0 < SPLAT_CMAX ? C - c0 : SPLAT_CMAX;
            if (nc == 4)
                hipLaunchKernelGGL(softsplat_tile_kernel<4>, dim3(n_tiles), dim3(256), 0, s, in, flow, out, ws->brange, guard,
                                   g_splat_cap, H, W, C, c0, nc, tiles_x, tiles_y);
            else
                hipLaunchKernelGGL(softsplat_tile_kernel<0>, dim3(n_tiles), dim3(256), 0, s, in, flow, out, ws->brange, guard,
                                   g_splat_cap, H, W, C, c0, nc, tiles_x, tiles_y);
        }
    }
    {
        TraceScope ts("splat_tail", s);
        hipLaunchKernelGGL(softsplat_tail_kernel, dim3(1024), dim3(256), 0, s, in, flow, out, ws->ctl, ws->ovf, g_splat_cap,
                           g_splat_mode == 0 ? 1 : 0, N, H, W, C, tiles_x, tiles_y);
    }
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---- cost volume -----------------------------------------------------------------------------------------
constexpr int CV_T = 16;            // output tile 16x16 pixels = 256 threads
constexpr int CV_TW = CV_T + 8;     // + 4-pixel halo each side
constexpr int CV_CK = 16;           // channels per LDS pass
constexpr int CV_S = CV_CK + 4;     // LDS pixel stride (floats)

// FULL: C is a multiple of the 16-channel LDS pass (M2M: 32): no per-element channel guards in the 81 x 16 inner loop (they were
// 1296 scalar branches per pass and cost 3/4 of the kernel's time)
template <bool FULL>
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
                    if (FULL || 4 * q + 0 < nc) a = __fadd_rn(a, fabsf(__fsub_rn(o[4 * q + 0], v.x)));
                    if (FULL || 4 * q + 1 < nc) a = __fadd_rn(a, fabsf(__fsub_rn(o[4 * q + 1], v.y)));
                    if (FULL || 4 * q + 2 < nc) a = __fadd_rn(a, fabsf(__fsub_rn(o[4 * q + 2], v.z)));
                    if (FULL || 4 * q + 3 < nc) a = __fadd_rn(a, fabsf(__fsub_rn(o[4 * q + 3], v.w)));
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
    if (C % CV_CK == 0)
        hipLaunchKernelGGL(costvol_kernel<true>, dim3(N * tiles_x * tiles_y), dim3(256), 0, s, one, one_cs, two, two_cs, two_swap, out, N,
                           H, W, C, out_cs, out_coff, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL(costvol_kernel<false>, dim3(N * tiles_x * tiles_y), dim3(256), 0, s, one, one_cs, two, two_cs, two_swap, out, N,
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
