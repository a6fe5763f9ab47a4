// M2M render as ONE kernel per timestep, and the tiled photometric kernel that feeds it (round 6).
//
//   m2m_photo_tiles_kernel  replaces m2m_photo_kernel for the C-side object: the same arithmetic (M2M_arch.py:945-1010, :559-561)
//                           per 32x32 tile, the partner image read as one float4 per tap from the compact plane img4, and — because the
//                           tile's refined flows tf_s are in registers anyway — the per-tile RANGES of every one of the 8 splat fields
//                           (what splat_blockrange computed per render, from a second pass over the scaled flows) and max |tf_s|.
//   m2m_render_kernel       replaces m2m_splat_inputs_kernel + the summation splat (flow_blockrange / softsplat_list / fallback / tail
//                           kernels) + m2m_combine_kernel: forwarp_mframe_mask of M2M_arch.py:551-581 with the splat of
//                           cupy_ops/softsplat.py:140-192 inside, hole fill :1026-1031, de-normalisation and crop :1033-1037.
//
// Why: the list-gather splat of m2m_ops.hip runs at 0.92 VALU-busy (profiles/r05_m2m_pmc_SQ_waves_waits.txt: 695 wave instructions per
// tile and splat, ~40 per contribution of which 15 are 64-bit address arithmetic and 12 the bilinear weights recomputed for every
// (pixel, source) pair) and moves 1.63x its algorithmic bytes (every output pixel re-reads flow and input of its sources from L2),
// between two kernels that only materialise its inputs (24 B per source) and re-read its outputs (16 B per pixel and splat).
// Here a workgroup owns a 32x32 tile of the FRAME and walks the 8 splats itself:
//   phase 1  every source of the tile's window (the rectangle of sources that can reach it, from the tile ranges scaled by the
//            timestep) is read ONCE with coalesced row loads — tf 8 B, e 4 B, image 16 B —, its splat input (img * td * e, td * e) and
//            its four bilinear weights are computed ONCE and staged in LDS (32 B), and it is filed under the cell of its north-west
//            target: one integer LDS atomic, a 2-byte entry;
//   phase 2  (only where a cell holds 3..6 sources) the cell's entries are sorted by source position;
//   phase 3  every thread gathers a 2x2 block of output pixels from the 3x3 cells around it: a cell's sources are loaded once from
//            LDS (two ds_read_b128) and serve up to four pixels; products and sums are packed (v_pk_mul_f32 / v_pk_add_f32: the same
//            IEEE operations, two channels per instruction), never fused — the reference's atomicAdd(out, in * w);
//   the combine (M2M_arch.py:569-581) runs in registers after every second splat; the frame is written once.
// The order of every sum is the list kernel's: per pixel south-east, south-west, north-east, north-west cell, ascending source
// position within a cell — so on windows that fit one strip the frame is BIT-IDENTICAL to the three-kernel path (tested), and to
// the sequential oracle for a uniform translation.
// There is no spill list, far pass or fallback launch: windows larger than the LDS stage are walked in strips, the reach of a
// window follows max |tf| (any displacement), a cell with more than 6 sources is gathered by scanning the strip — slow, exact, rare.
#include <atomic>

#include <map>
#include <mutex>

#include "vfi_common.h"
#include "m2m_warp.h"

#include "../../include/vfi_hip.h"

namespace vfi {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int RT = 32;                 // tile edge
constexpr int RCW = RT + 1;            // cells per row: north-west targets lx in [-1, 31]
constexpr int RCELLS = RCW * RCW;
constexpr int RK = 6;                  // entries a cell holds (uint4: count | 6 x u16 byte offsets into the stage)
constexpr int RCAP = 1824;             // sources staged per strip (42.7 x 42.7; entry = index * 32 fits 16 bits; 2 workgroups of 79.6 KB per CU)

struct RSrc {
    float4 in;      // (img * td) * e, td * e                      (M2M_arch.py:1012-1024, :563-567)
    float4 w;       // bilinear weights nw, ne, sw, se of the target (softsplat.py:160-173)
};

static unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

// ---- photometric metric per tile + tile ranges -----------------------------------------------------------------------------------
// grid: 2 * tiles (direction d, tile ty, tx); thread (tid & 31, tid >> 5) walks 4 rows.  Arithmetic: m2m_photo_kernel's, verbatim.
// brange[s][tile] = [fx_min, fx_max, fy_min, fy_max] over the tile's FINITE refined flows tf_s (inverted when none), s = 2 b + d;
// smax_bits[s] = max |tf_s|_inf over the finite ones (float bits; zeroed before the launch).
__global__ __launch_bounds__(256) void m2m_photo_tiles_kernel(const float* __restrict__ d0, int d0_cs, const float4* __restrict__ img4,
                                                              const float* __restrict__ r, int r_cs, float alpha, float* __restrict__ TF,
                                                              float* __restrict__ E, float4* __restrict__ brange,
                                                              unsigned* __restrict__ smax_bits, int H, int W, int tiles_x, int tiles_y,
                                                              float stepx, float stepy, float sclx, float scly) {
    const int tid = threadIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int d = blockIdx.x / tiles;
    const int trem = blockIdx.x - d * tiles;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const long hw = (long)H * W;
    const int x = tx * RT + (tid & 31);
    const float big = 3.0e38f;
    float lo_x[4], hi_x[4], lo_y[4], hi_y[4], amax[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) lo_x[b] = big, hi_x[b] = -big, lo_y[b] = big, hi_y[b] = -big, amax[b] = 0.f;
    const float4* other = img4 + (size_t)(d ^ 1) * hw;
    const bool vec_r = (r_cs & 3) == 0 && ((uintptr_t)r & 15) == 0;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int y = ty * RT + (tid >> 5) + 8 * q;
        if (x >= W || y >= H) continue;
        const long p = (long)y * W + x;
        const long idx = d * hw + p;
        const float* me = d0 + (size_t)idx * d0_cs;
        const float* rp = r + (size_t)idx * r_cs;
        // the 9 values of r as wide loads where the layout allows (the object's r: 12 floats per pixel, 16-byte aligned): 3 loads instead
        // of 9 scalar ones at a 48-byte lane stride
        float rr[9];
        if (vec_r) {
            const float4 q0 = *(const float4*)rp, q1 = *(const float4*)(rp + 4);
            rr[0] = q0.x, rr[1] = q0.y, rr[2] = q0.z, rr[3] = q0.w, rr[4] = q1.x, rr[5] = q1.y, rr[6] = q1.z, rr[7] = q1.w, rr[8] = rp[8];
        } else {
#pragma unroll
            for (int j = 0; j < 9; ++j) rr[j] = rp[j];
        }
        const float wei = __fadd_rn(__fmul_rn(1.0f / (1.0f + expf(-rr[8])), 0.8f), 0.1f);
        const float4 im = img4[(size_t)idx];
        const float2 f01 = *(const float2*)me;      // d0_cs is even and d0 8-byte aligned (checked by the launcher)
        const float f0 = f01.x, f1 = f01.y;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float fx = __fadd_rn(f0, rr[2 * b]), fy = __fadd_rn(f1, rr[2 * b + 1]);
            const WarpTap t = m2m_taps(x, y, fx, fy, H, W, stepx, stepy, sclx, scly);
            float w0 = 0.f, w1 = 0.f, w2 = 0.f;      // torch accumulates nw, ne, sw, se in that order starting from 0 (tap_acc)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t.off[k] >= 0) {
                    const float4 v = other[t.off[k]];
                    w0 = __fadd_rn(w0, __fmul_rn(v.x, t.w[k]));
                    w1 = __fadd_rn(w1, __fmul_rn(v.y, t.w[k]));
                    w2 = __fadd_rn(w2, __fmul_rn(v.z, t.w[k]));
                }
            const float a0 = fabsf(__fsub_rn(im.x, w0)), a1 = fabsf(__fsub_rn(im.y, w1)), a2 = fabsf(__fsub_rn(im.z, w2));
            const float m = __fdiv_rn(__fadd_rn(__fadd_rn(a0, a1), a2), 3.0f);
            float ph = fmaxf(__fsub_rn(1.0f, __fmul_rn(wei, m)), 0.001f);
            ph = __fmul_rn(ph, ph);
            const float met = fminf(fmaxf(__fmul_rn(alpha, ph), -20.0f), 20.0f);
            const size_t s = (size_t)(2 * b + d) * hw + p;
            *(float2*)(TF + s * 2) = make_float2(fx, fy);
            E[s] = expf(met);
            if (isfinite(fx) && isfinite(fy)) {
                lo_x[b] = fminf(lo_x[b], fx), hi_x[b] = fmaxf(hi_x[b], fx);
                lo_y[b] = fminf(lo_y[b], fy), hi_y[b] = fmaxf(hi_y[b], fy);
                amax[b] = fmaxf(amax[b], fmaxf(fabsf(fx), fabsf(fy)));
            }
        }
    }
    __shared__ float red[4][4][5];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        for (int o = 32; o > 0; o >>= 1) {
            lo_x[b] = fminf(lo_x[b], __shfl_xor(lo_x[b], o)), hi_x[b] = fmaxf(hi_x[b], __shfl_xor(hi_x[b], o));
            lo_y[b] = fminf(lo_y[b], __shfl_xor(lo_y[b], o)), hi_y[b] = fmaxf(hi_y[b], __shfl_xor(hi_y[b], o));
            amax[b] = fmaxf(amax[b], __shfl_xor(amax[b], o));
        }
        if ((tid & 63) == 0) {
            float* q = red[tid >> 6][b];
            q[0] = lo_x[b], q[1] = hi_x[b], q[2] = lo_y[b], q[3] = hi_y[b], q[4] = amax[b];
        }
    }
    __syncthreads();
    if (tid < 4) {
        const int b = tid;
        float v[5];
        for (int j = 0; j < 5; ++j) {
            const float a0 = red[0][b][j], a1 = red[1][b][j], a2 = red[2][b][j], a3 = red[3][b][j];
            v[j] = (j == 0 || j == 2) ? fminf(fminf(a0, a1), fminf(a2, a3)) : fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
        }
        const int s = 2 * b + d;
        brange[(size_t)s * tiles + trem] = make_float4(v[0], v[1], v[2], v[3]);
        // non-negative floats order as their bit patterns; a plain look first keeps the workgroups from queueing on one address
        if (v[4] > 0.f && __float_as_uint(v[4]) > __builtin_nontemporal_load(&smax_bits[s])) atomicMax(&smax_bits[s], __float_as_uint(v[4]));
    }
}

// ---- the render ------------------------------------------------------------------------------------------------------------------
struct RenderArgs {
    const float4* img4;      // [2][Hp*Wp] normalised image, w = 1
    const float* TF;         // [8][Hp*Wp][2]
    const float* E;          // [8][Hp*Wp]
    const float4* brange;    // [8][tiles]
    const unsigned* smax;    // [8] float bits
    const float* stats;      // mean, std + 1e-7
    float* out;              // [H][W][3]
    float t;
    int Hp, Wp, H, W, tiles_x, tiles_y;
};

struct RWin {
    int x0, x1, y0, y1;
};

// the window of splat s for this tile: the bounding box, over the source blocks within reach, of the block's pixels inside
// (tile - [f_min, f_max] * tm) — m2m_ops.hip's tile_window with the ranges scaled by the timestep (tm >= 0: scaling is monotone) and
// the reach taken from max |tf_s| * tm instead of a fixed two blocks.  All 256 threads; red = 16 ints of LDS.
__device__ static inline RWin render_window(const float4* __restrict__ br, float reach, float tm, int ty, int tx, int tiles_x, int tiles_y,
                                            int H, int W, int* red) {
    const int tid = threadIdx.x;
    const float nbf = fminf(ceilf((reach + 1.0f) * (1.0f / 32.0f)), (float)(tiles_x > tiles_y ? tiles_x : tiles_y));
    const int nb = nbf >= 1.0f ? (int)nbf : 1;
    const int side = 2 * nb + 1, total = side * side;
    int wx0 = 1 << 30, wx1 = -(1 << 30), wy0 = 1 << 30, wy1 = -(1 << 30);
    const int X0 = tx * RT, Y0 = ty * RT;
    for (int i = tid; i < total; i += 256) {
        const int dy = i / side - nb, dx = i % side - nb;
        const int by = ty + dy, bx = tx + dx;
        if (by < 0 || by >= tiles_y || bx < 0 || bx >= tiles_x) continue;
        const float4 r = br[by * tiles_x + bx];
        if (!(r.x <= r.y && r.z <= r.w)) continue;
        const float lim = 1.0e7f;
        const float fx0 = fminf(fmaxf(r.x * tm, -lim), lim), fx1 = fminf(fmaxf(r.y * tm, -lim), lim);
        const float fy0 = fminf(fmaxf(r.z * tm, -lim), lim), fy1 = fminf(fmaxf(r.w * tm, -lim), lim);
        // source s reaches the tile iff X0 - 1 <= s + f < X0 + RT: s in [ceil(X0 - 1 - f_max), ceil(X0 + RT - f_min)) — floor / ceil of the
        // bounds moved OUT by eps (the float sums s + f and these differences round: an ulp of 2000 + |f|), i.e. a pixel of margin only where a bound lies within eps of an integer
        const float ex = 0.02f + 1e-6f * fmaxf(fabsf(fx0), fabsf(fx1)), ey = 0.02f + 1e-6f * fmaxf(fabsf(fy0), fabsf(fy1));
        const int lx = max((int)ceilf((float)(X0 - 1) - fx1 - ex), bx * RT);
        const int hx = min((int)ceilf((float)(X0 + RT) - fx0 + ex), min(bx * RT + RT, W));
        const int ly = max((int)ceilf((float)(Y0 - 1) - fy1 - ey), by * RT);
        const int hy = min((int)ceilf((float)(Y0 + RT) - fy0 + ey), min(by * RT + RT, H));
        if (lx < hx && ly < hy) wx0 = min(wx0, lx), wx1 = max(wx1, hx), wy0 = min(wy0, ly), wy1 = max(wy1, hy);
    }
    for (int o = 32; o > 0; o >>= 1) {
        wx0 = min(wx0, __shfl_xor(wx0, o)), wx1 = max(wx1, __shfl_xor(wx1, o));
        wy0 = min(wy0, __shfl_xor(wy0, o)), wy1 = max(wy1, __shfl_xor(wy1, o));
    }
    __syncthreads();      // red is reused from the previous splat
    if ((tid & 63) == 0) {
        int* q = red + 4 * (tid >> 6);
        q[0] = wx0, q[1] = wx1, q[2] = wy0, q[3] = wy1;
    }
    __syncthreads();
    RWin w;
    w.x0 = min(min(red[0], red[4]), min(red[8], red[12]));
    w.x1 = max(max(red[1], red[5]), max(red[9], red[13]));
    w.y0 = min(min(red[2], red[6]), min(red[10], red[14]));
    w.y1 = max(max(red[3], red[7]), max(red[11], red[15]));
    if (w.x1 <= w.x0 || w.y1 <= w.y0) w = RWin{0, 0, 0, 0};
    return w;
}

// acc += in * w, unfused, two channels per instruction
__device__ static inline void racc(v2f& a01, v2f& a23, const float4& in, float w) {
#pragma clang fp contract(off)
    const v2f w2 = {w, w};
    const v2f i01 = {in.x, in.y}, i23 = {in.z, in.w};
    const v2f p01 = i01 * w2, p23 = i23 * w2;
    a01 = a01 + p01;
    a23 = a23 + p23;
}

// phases 2 and 3 of one strip: order the crowded cells, then every thread gathers its 2x2 pixel block from the 3x3 cells around it.
// n = sources staged; cells / scell / flags as phase 1 left them (after a barrier).  Ends WITHOUT a barrier.
__device__ __forceinline__ void render_gather(unsigned char* lds_raw, uint4* cells, const unsigned short* scell, const int* flags, int n,
                                              int cbase, v2f (&cur01)[4], v2f (&cur23)[4]) {
    const int tid = threadIdx.x;
    const RSrc* const stage = (const RSrc*)lds_raw;
    // ---- phase 2: cells with 3..6 sources: ascending source position (two are ordered when read; more than 6 are scanned)
    if (flags[0] | flags[1]) {
        for (int c = tid; c < RCELLS; c += 256) {
            const unsigned cnt = cells[c].x;
            if (cnt < 3u) continue;
            unsigned short* l = (unsigned short*)&cells[c] + 2;
            const int k = (int)min(cnt, (unsigned)RK);
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
    }
    const bool overflowed = flags[1] != 0;
    unsigned ovmask = 0;      // bit cj * 3 + ci: that cell of this thread's 3x3 holds more sources than a cell's list
    // ---- phase 3: gather the 2x2 block from its 3x3 cells, in raster order of the cells (= SE, SW, NE, NW per pixel)
#pragma unroll
    for (int cj = 0; cj < 3; ++cj) {
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const int cidx = cbase + cj * RCW + ci;
            const uint4 c = cells[cidx];
            const unsigned cnt = c.x;
            if (__builtin_expect(overflowed, 0) && cnt > (unsigned)RK) {
                ovmask |= 1u << (cj * 3 + ci);
                continue;
            }
            unsigned e[RK] = {c.y & 0xffffu, c.y >> 16, c.z & 0xffffu, c.z >> 16, c.w & 0xffffu, c.w >> 16};
            if (cnt == 2u) {
                const unsigned lo = min(e[0], e[1]), hi = max(e[0], e[1]);
                e[0] = lo, e[1] = hi;
            }
#pragma unroll
            for (int q = 0; q < RK; ++q) {
                const bool live = (unsigned)q < cnt;
                if (!__any(live)) break;
                if (live) {
                    const RSrc v = *(const RSrc*)(lds_raw + e[q]);
#pragma unroll
                    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                        for (int pa = 0; pa < 2; ++pa)
                            if (ci - pa >= 0 && ci - pa <= 1 && cj - pb >= 0 && cj - pb <= 1) {
                                const int k = (pa + 1 - ci) + 2 * (pb + 1 - cj);
                                racc(cur01[2 * pb + pa], cur23[2 * pb + pa], v.in, k == 0 ? v.w.x : (k == 1 ? v.w.y : (k == 2 ? v.w.z : v.w.w)));
                            }
                }
            }
        }
    }
    // ---- cells with more sources than a list holds (strongly convergent flow): ONE pass over the strip's sources, in source order,
    // takes every source filed under one of this thread's overflowed cells (their sums follow the listed cells': a fixed order still)
    if (__builtin_expect(overflowed, 0) && __any(ovmask != 0u)) {
        if (ovmask != 0u) {
            for (int j = 0; j < n; ++j) {
                const int d = (int)scell[j] - cbase;
                if ((unsigned)d > (unsigned)(2 * RCW + 2)) continue;
                const int cj = d >= 2 * RCW ? 2 : (d >= RCW ? 1 : 0);
                const int ci = d - cj * RCW;
                if (ci > 2 || !((ovmask >> (cj * 3 + ci)) & 1u)) continue;
                const RSrc v = stage[j];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int pa = 0; pa < 2; ++pa) {
                        const int ax = pa + 1 - ci, ay = pb + 1 - cj;      // corner of the source this pixel is: 0 / 1 each, else not covered
                        if ((unsigned)ax <= 1u && (unsigned)ay <= 1u) {
                            const float wx0 = ay ? v.w.z : v.w.x, wx1 = ay ? v.w.w : v.w.y;
                            racc(cur01[2 * pb + pa], cur23[2 * pb + pa], v.in, ax ? wx1 : wx0);
                        }
                    }
            }
        }
    }
}

__global__ __launch_bounds__(256) void m2m_render_kernel(const RenderArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    RSrc* const stage = (RSrc*)lds_raw;                                          // RCAP x 32 B
    uint4* const cells = (uint4*)(lds_raw + sizeof(RSrc) * RCAP);                // RCELLS x 16 B: count | 6 x u16 entries
    unsigned short* const scell = (unsigned short*)(cells + RCELLS);             // RCAP: cell of every staged source (0xffff: none)
    int* const red = (int*)(scell + RCAP);                                       // 16 ints (window) + flags
    int* const flags = red + 16;                                                 // [0]: a cell reached 3 entries, [1]: a cell overflowed
    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (each XCD has its own L2), so XCD x takes the contiguous band of tiles
    // [x * per, (x + 1) * per) — neighbouring tiles, whose windows overlap by the halo and share cache lines, then share an L2.  With the
    // plain order every source line was fetched from HBM by two XCDs (r6 PMC: 708 MB fetched for 267 MB of inputs).
    const int per_xcd = (int)gridDim.x >> 3;
    const int tile = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (tile >= a.tiles_x * a.tiles_y) return;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int X0 = tx * RT, Y0 = ty * RT;
    if (X0 >= a.W || Y0 >= a.H) return;      // a tile of the padding only: no output pixel (the whole workgroup leaves)
    const int tiles = a.tiles_x * a.tiles_y;
    const long hw = (long)a.Hp * a.Wp;
    const float t = a.t, t1 = __fsub_rn(1.0f, t);
    // phase-3 ownership: thread (bx, by) gathers the 2x2 pixel block at (2 bx, 2 by) of the tile
    const int pbx = tid & 15, pby = tid >> 4;
    const int cbase = (2 * pby) * RCW + 2 * pbx;      // cell (ci, cj) of the block = cbase + cj * RCW + ci; pixel (a, b) takes corner
                                                      // k = (a + 1 - ci) + 2 (b + 1 - cj) of the sources filed under it
    v2f acc01[4], acc2n[4];      // per pixel pi = 2 b + a: (sum r, sum g), (sum b, norm)      [M2M_arch.py:569-581]
    v2f ev01[4], ev23[4];        // the even splat of the current branch
#pragma unroll
    for (int i = 0; i < 4; ++i) acc01[i] = v2f{0.f, 0.f}, acc2n[i] = v2f{0.f, 0.f};

#pragma unroll 1
    for (int s = 0; s < 8; ++s) {
        const int d = s & 1;
        const float td = d ? t : t1, tm = d ? t1 : t;
        const float reach = __uint_as_float(a.smax[s]) * tm;
        const RWin win = render_window(a.brange + (size_t)s * tiles, reach, tm, ty, tx, a.tiles_x, a.tiles_y, a.Hp, a.Wp, red);
        const int ww = win.x1 - win.x0, wh = win.y1 - win.y0;
        const long total = (long)ww * wh;
        const float2* const TFs = (const float2*)a.TF + (size_t)s * hw;
        const float* const Es = a.E + (size_t)s * hw;
        const float4* const Is = a.img4 + (size_t)d * hw;
        v2f cur01[4], cur23[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) cur01[i] = v2f{0.f, 0.f}, cur23[i] = v2f{0.f, 0.f};
        // 256 consecutive window indices advance (dy, dx) by (q256, r256)
        const int q256 = ww > 0 ? 256 / ww : 0, r256 = ww > 0 ? 256 - q256 * ww : 0;
#pragma unroll 1
        for (long i0 = 0; i0 < total; i0 += RCAP) {
            const int n = (int)min((long)RCAP, total - i0);
            // ---- clear the cells
            __syncthreads();      // the previous strip's gather is done with cells and stage
            for (int i = tid; i < RCELLS; i += 256) cells[i] = make_uint4(0u, 0u, 0u, 0u);
            if (tid < 2) flags[tid] = 0;
            __syncthreads();
            // ---- phase 1: stage every source of the strip, file it under its north-west target cell (4 sources per thread in flight)
            {
                constexpr int U = 4;
                const long first = i0 + tid;
                int dy = (int)(first / ww), dx = (int)(first - (long)dy * ww);
                for (int j0 = tid; j0 < n; j0 += 256 * U) {
                    float2 tfv[U];
                    float ev[U];
                    float4 imv[U];
                    int sxs[U], sys[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const bool in = j0 + 256 * u < n;
                        sxs[u] = win.x0 + dx, sys[u] = win.y0 + dy;
                        const int p = in ? sys[u] * a.Wp + sxs[u] : win.y0 * a.Wp + win.x0;
                        tfv[u] = TFs[p];
                        ev[u] = Es[p];
                        imv[u] = Is[p];
                        dx += r256, dy += q256;
                        if (dx >= ww) dx -= ww, dy += 1;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int j = j0 + 256 * u;
                        if (j >= n) break;
                        const float2 tf = tfv[u];
                        const float e = ev[u];
                        const float4 im = imv[u];
                        const float flx = __fmul_rn(tf.x, tm), fly = __fmul_rn(tf.y, tm);
                        const float fx = __fadd_rn((float)sxs[u], flx), fy = __fadd_rn((float)sys[u], fly);
                        const float x0f = floorf(fx), y0f = floorf(fy);
                        const float x1f = __fadd_rn(x0f, 1.0f), y1f = __fadd_rn(y0f, 1.0f);      // == (float)(x0 + 1) for every |x0| < 2^24
                        const float bx = __fsub_rn(x1f, fx), ax = __fsub_rn(fx, x0f), by = __fsub_rn(y1f, fy), ay = __fsub_rn(fy, y0f);
                        RSrc v;
                        v.w = make_float4(__fmul_rn(bx, by), __fmul_rn(ax, by), __fmul_rn(bx, ay), __fmul_rn(ax, ay));
                        v.in = make_float4(__fmul_rn(__fmul_rn(im.x, td), e), __fmul_rn(__fmul_rn(im.y, td), e), __fmul_rn(__fmul_rn(im.z, td), e),
                                           __fmul_rn(td, e));
                        stage[j] = v;
                        // softsplat.py:157-158: non-finite targets never splat (fx - fx is 0 only for a finite fx); targets beyond this
                        // tile's cells belong to another tile.  The float compares come first: they keep the int conversion defined.
                        const float lxf = x0f - (float)(X0 - 1), lyf = y0f - (float)(Y0 - 1);
                        const bool ok = (fx - fx == 0.f) && (fy - fy == 0.f) && lxf >= 0.f && lxf <= (float)RT && lyf >= 0.f && lyf <= (float)RT;
                        int cell = 0xffff;
                        if (ok) {
                            cell = (int)lyf * RCW + (int)lxf;
                            const unsigned slot = atomicAdd(&cells[cell].x, 1u);
                            if (slot < (unsigned)RK) ((unsigned short*)&cells[cell])[2 + slot] = (unsigned short)(j * (int)sizeof(RSrc));
                            if (slot >= 2u) flags[slot >= (unsigned)RK ? 1 : 0] = 1;
                        }
                        scell[j] = (unsigned short)cell;
                    }
                }
            }
            __syncthreads();
            render_gather(lds_raw, cells, scell, flags, n, cbase, cur01, cur23);
        }
        // ---- forwarp_mframe_mask's accumulation (M2M_arch.py:569-581): per branch (fwd + bwd), norm += (fwd.w + 1e-7) + (bwd.w + 1e-7)
        if (!d) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ev01[i] = cur01[i], ev23[i] = cur23[i];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc01[i] = acc01[i] + (ev01[i] + cur01[i]);
                const float s2 = __fadd_rn(ev23[i].x, cur23[i].x);
                const float nn = __fadd_rn(__fadd_rn(ev23[i].y, 0.0000001f), __fadd_rn(cur23[i].y, 0.0000001f));
                acc2n[i] = acc2n[i] + v2f{s2, nn};
            }
        }
    }
    // ---- hole fill (:1026-1031), de-normalisation and crop (:1033-1037)
    const float mean = a.stats[0], sd = a.stats[1];
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int x = X0 + 2 * pbx + pa, y = Y0 + 2 * pby + pb;
            if (x >= a.W || y >= a.H) continue;
            const int i = 2 * pb + pa;
            const long p = (long)y * a.Wp + x;
            const float norm = acc2n[i].y;
            const bool hole = norm < 0.00001f;
            const float4 ia = a.img4[p], ib = a.img4[hw + p];
            const float accv[3] = {acc01[i].x, acc01[i].y, acc2n[i].x};
            const float va[3] = {ia.x, ia.y, ia.z}, vb[3] = {ib.x, ib.y, ib.z};
            float* o = a.out + ((size_t)y * a.W + x) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = __fdiv_rn(accv[c], norm);
                if (hole) v = __fadd_rn(v, __fadd_rn(__fmul_rn(t1, va[c]), __fmul_rn(t, vb[c])));
                o[c] = __fadd_rn(__fmul_rn(v, sd), mean);
            }
        }
}

// ---- the generic 4-channel summation splat on the same machinery --------------------------------------------------------------------
// vfi_softsplat_sum for C == 4 (softsplat_out, cupy_ops/softsplat.py:140-192): one splat, input and flow as given.  Differences from
// the M2M kernel above, both for INCOHERENT fields (i.i.d. N(0, 8 px) — a tile's window holds 7x as many sources
// as land in it):
//   * sources are CLASSIFIED first (flow load, target cell, 18 instructions) in chunks of 1024, and only those that reach the tile are
//     compacted — by ballot / popcount ranks in raster order, so the stage order stays the source order — into an index list;
//   * the expensive part (input load, weights, staging, filing) then runs densely over that list, and a strip is flushed (gathered)
//     only when the list is full: an 86 x 86 window of which 1100 sources land is ONE gather, not five.
constexpr int GCAP = 1664;             // staged sources per flush (2 workgroups of 79.9 KB per CU)
constexpr int GU = 6;                  // sources classified per thread and step: 6 flow loads in flight (a whole chunk must fit the stage)
constexpr int GCHUNK = 256 * GU;       // sources classified per step
static_assert(GCHUNK <= GCAP, "a chunk whose sources all reach the tile must fit the stage");

struct Splat4Args {
    const float4* in;        // [N][H*W]
    const float2* flow;      // [N][H*W]
    float4* out;             // [N][H*W]
    const float4* brange;    // [N][tiles]
    const unsigned* smax;    // [N] float bits
    int H, W, tiles_x, tiles_y;
};

// per 32x32 tile: range of the finite flows, and max |f| per image (zeroed before the launch)
__global__ __launch_bounds__(256) void flow_tile_ranges_kernel(const float2* __restrict__ flow, int H, int W, int tiles_x, int tiles_y,
                                                               float4* __restrict__ brange, unsigned* __restrict__ smax_bits) {
    const int tid = threadIdx.x;
    const int tiles = tiles_x * tiles_y;
    const int n = blockIdx.x / tiles;
    const int trem = blockIdx.x - n * tiles;
    const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
    const int x = tx * RT + (tid & 31);
    const float big = 3.0e38f;
    float x0 = big, x1 = -big, y0 = big, y1 = -big, am = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int y = ty * RT + (tid >> 5) + 8 * q;
        if (x < W && y < H) {
            const float2 f = flow[((size_t)n * H + y) * W + x];
            if (isfinite(f.x) && isfinite(f.y)) {
                x0 = fminf(x0, f.x), x1 = fmaxf(x1, f.x), y0 = fminf(y0, f.y), y1 = fmaxf(y1, f.y);
                am = fmaxf(am, fmaxf(fabsf(f.x), fabsf(f.y)));
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o)), x1 = fmaxf(x1, __shfl_xor(x1, o));
        y0 = fminf(y0, __shfl_xor(y0, o)), y1 = fmaxf(y1, __shfl_xor(y1, o));
        am = fmaxf(am, __shfl_xor(am, o));
    }
    __shared__ float wr[4][5];
    if ((tid & 63) == 0) {
        float* r = wr[tid >> 6];
        r[0] = x0, r[1] = x1, r[2] = y0, r[3] = y1, r[4] = am;
    }
    __syncthreads();
    if (tid == 0) {
        brange[blockIdx.x] = make_float4(fminf(fminf(wr[0][0], wr[1][0]), fminf(wr[2][0], wr[3][0])), fmaxf(fmaxf(wr[0][1], wr[1][1]), fmaxf(wr[2][1], wr[3][1])),
                                         fminf(fminf(wr[0][2], wr[1][2]), fminf(wr[2][2], wr[3][2])), fmaxf(fmaxf(wr[0][3], wr[1][3]), fmaxf(wr[2][3], wr[3][3])));
        const float a = fmaxf(fmaxf(wr[0][4], wr[1][4]), fmaxf(wr[2][4], wr[3][4]));
        // (a plain look first: 2040 workgroups queueing on one address cost 20 us; after the first few, hardly any still raises the maximum)
        if (a > 0.f && __float_as_uint(a) > __builtin_nontemporal_load(&smax_bits[n])) atomicMax(&smax_bits[n], __float_as_uint(a));
    }
}

__global__ __launch_bounds__(256) void softsplat4_kernel(const Splat4Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    RSrc* const stage = (RSrc*)lds_raw;                                          // GCAP x 32 B
    uint4* const cells = (uint4*)(lds_raw + sizeof(RSrc) * GCAP);                // RCELLS x 16 B
    unsigned short* const scell = (unsigned short*)(cells + RCELLS);             // GCAP
    unsigned* const sidx = (unsigned*)(scell + GCAP);                            // GCAP window indices of the sources that reach the tile
    int* const red = (int*)(sidx + GCAP);                                        // 16 ints (window)
    int* const flags = red + 16;                                                 // 2
    int* const wcnt = flags + 2;                                                 // [2][GU][4 waves] classification counts, double-buffered
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = a.tiles_x * a.tiles_y;
    const int n = blockIdx.x / tiles;
    const int trem = blockIdx.x - n * tiles;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int X0 = tx * RT, Y0 = ty * RT;
    const long hw = (long)a.H * a.W;
    const float4* const Is = a.in + (size_t)n * hw;
    const float2* const Fs = a.flow + (size_t)n * hw;
    const int pbx = tid & 15, pby = tid >> 4;
    const int cbase = (2 * pby) * RCW + 2 * pbx;
    v2f cur01[4], cur23[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur01[i] = v2f{0.f, 0.f}, cur23[i] = v2f{0.f, 0.f};
    const RWin win = render_window(a.brange + (size_t)n * tiles, __uint_as_float(a.smax[n]), 1.0f, ty, tx, a.tiles_x, a.tiles_y, a.H, a.W, red);
    const int ww = win.x1 - win.x0, wh = win.y1 - win.y0;
    const long total = (long)ww * wh;
    const float inv_ww = ww > 0 ? 1.0f / (float)ww : 0.f;
    const bool small = total < (1L << 22);      // window indices exact in fp32: the quotient by reciprocal + one fix-up
    for (int i = tid; i < RCELLS; i += 256) cells[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 2) flags[tid] = 0;
    int staged = 0;

    // stage + file the listed sources, gather them, clear the cells; all threads
    auto flush = [&](int ns) {
        __syncthreads();      // the index list (and the cleared cells) are visible
        for (int k0 = tid; k0 < ns; k0 += 1024) {
            float2 fv[4];
            float4 iv[4];
            int sxs[4], sys[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {      // 8 loads in flight per thread
                const int k = k0 + 256 * u;
                const unsigned i = sidx[k < ns ? k : tid];
                int dy;
                if (small) {
                    dy = (int)((float)i * inv_ww);
                    const int r = (int)i - dy * ww;
                    dy += r >= ww ? 1 : (r < 0 ? -1 : 0);
                } else {
                    dy = (int)(i / (unsigned)ww);
                }
                const int dx = (int)i - dy * ww;
                sxs[u] = win.x0 + dx, sys[u] = win.y0 + dy;
                const int p = sys[u] * a.W + sxs[u];
                fv[u] = Fs[p];
                iv[u] = Is[p];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 256 * u;
                if (k >= ns) break;
                const float2 f = fv[u];
                const float fx = __fadd_rn((float)sxs[u], f.x), fy = __fadd_rn((float)sys[u], f.y);
                const float x0f = floorf(fx), y0f = floorf(fy);
                const float x1f = __fadd_rn(x0f, 1.0f), y1f = __fadd_rn(y0f, 1.0f);
                const float bx = __fsub_rn(x1f, fx), ax = __fsub_rn(fx, x0f), by = __fsub_rn(y1f, fy), ay = __fsub_rn(fy, y0f);
                RSrc v;
                v.w = make_float4(__fmul_rn(bx, by), __fmul_rn(ax, by), __fmul_rn(bx, ay), __fmul_rn(ax, ay));
                v.in = iv[u];
                stage[k] = v;
                const int cell = (int)(y0f - (float)(Y0 - 1)) * RCW + (int)(x0f - (float)(X0 - 1));      // in range: the classification kept only those
                const unsigned slot = atomicAdd(&cells[cell].x, 1u);
                if (slot < (unsigned)RK) ((unsigned short*)&cells[cell])[2 + slot] = (unsigned short)(k * (int)sizeof(RSrc));
                if (slot >= 2u) flags[slot >= (unsigned)RK ? 1 : 0] = 1;
                scell[k] = (unsigned short)cell;
            }
        }
        __syncthreads();
        render_gather(lds_raw, cells, scell, flags, ns, cbase, cur01, cur23);
        __syncthreads();
        for (int i = tid; i < RCELLS; i += 256) cells[i] = make_uint4(0u, 0u, 0u, 0u);
        if (tid < 2) flags[tid] = 0;
    };

    int par = 0;
#pragma unroll 1
    for (long c0 = 0; c0 < total; c0 += GCHUNK, par ^= 1) {
        // ---- classify 4 sources per thread: does the source's north-west target fall on this tile's cells?
        const long first = c0 + tid;
        int dy = (int)(first / ww), dx = (int)(first - (long)dy * ww);
        const int q256 = 256 / ww, r256 = 256 - q256 * ww;
        bool ok[GU];
        unsigned rank[GU];
        float2 fv[GU];
        int sxs[GU], sys[GU];
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const bool in = first + 256 * u < total;
            sxs[u] = win.x0 + dx, sys[u] = win.y0 + dy;
            fv[u] = Fs[in ? sys[u] * a.W + sxs[u] : win.y0 * a.W + win.x0];
            dx += r256, dy += q256;
            if (dx >= ww) dx -= ww, dy += 1;
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const bool in = first + 256 * u < total;
            const float fx = __fadd_rn((float)sxs[u], fv[u].x), fy = __fadd_rn((float)sys[u], fv[u].y);
            const float lxf = floorf(fx) - (float)(X0 - 1), lyf = floorf(fy) - (float)(Y0 - 1);
            // softsplat.py:157-158: non-finite targets never splat (fx - fx is 0 only for a finite fx)
            ok[u] = in && (fx - fx == 0.f) && (fy - fy == 0.f) && lxf >= 0.f && lxf <= (float)RT && lyf >= 0.f && lyf <= (float)RT;
            const unsigned long long b = __ballot(ok[u]);
            rank[u] = (unsigned)__popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[par * 4 * GU + u * 4 + wave] = (int)__popcll(b);
        }
        __syncthreads();
        int base[GU], sum = 0;      // raster order of a chunk: u major, then wave, then lane
#pragma unroll
        for (int u = 0; u < GU; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (w == wave) base[u] = sum;
                sum += wcnt[par * 4 * GU + u * 4 + w];
            }
        if (staged + sum > GCAP) {      // uniform: the list would overflow — gather what is staged first (a chunk alone always fits)
            flush(staged);
            staged = 0;
        }
#pragma unroll
        for (int u = 0; u < GU; ++u)
            if (ok[u]) sidx[staged + base[u] + (int)rank[u]] = (unsigned)(first + 256 * u);
        staged += sum;
    }
    if (staged > 0) flush(staged);
    // ---- the tile's 2x2 blocks out
#pragma unroll
    for (int pb = 0; pb < 2; ++pb)
#pragma unroll
        for (int pa = 0; pa < 2; ++pa) {
            const int x = X0 + 2 * pbx + pa, y = Y0 + 2 * pby + pb;
            if (x >= a.W || y >= a.H) continue;
            const int i = 2 * pb + pa;
            a.out[(size_t)n * hw + (size_t)y * a.W + x] = make_float4(cur01[i].x, cur01[i].y, cur23[i].x, cur23[i].y);
        }
}

constexpr size_t kSplat4Lds = sizeof(RSrc) * GCAP + sizeof(uint4) * RCELLS + sizeof(unsigned short) * GCAP + sizeof(unsigned) * GCAP + sizeof(int) * (16 + 2 + 2 * 4 * GU + 14);

constexpr size_t kRenderLds = sizeof(RSrc) * RCAP + sizeof(uint4) * RCELLS + sizeof(unsigned short) * RCAP + sizeof(int) * 32;

// compact image plane [2][Hp*Wp][4] = (normalised r, g, b, 1) from d0's channels 2..4
__global__ void m2m_img4_kernel(const float* __restrict__ d0, int d0_cs, float4* __restrict__ img4, long n) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float* s = d0 + (size_t)idx * d0_cs + 2;
    img4[idx] = make_float4(s[0], s[1], s[2], 1.0f);
}

// backwarp of the PARTNER image by this image's flow (MotionRefineNet.forward's warped image, M2M_arch.py:866-890 through backwarp
// :24-92): warp_m2m_kernel<false>'s arithmetic for C = 3 with one float4 load per tap from the compact plane instead of three scalar
// loads at a 32-byte stride.
__global__ void m2m_warp_img4_kernel(const float4* __restrict__ img4, const float* __restrict__ flow, int flow_cs, float* __restrict__ out,
                                     int out_cs, int H, int W, float stepx, float stepy, float sclx, float scly) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)H * W;
    if (idx >= 2 * hw) return;
    const int n = idx >= hw;
    const long p = idx - n * hw;
    const int x = p % W, y = p / W;
    const WarpTap t = m2m_taps(x, y, flow[idx * flow_cs], flow[idx * flow_cs + 1], H, W, stepx, stepy, sclx, scly);
    const float4* b = img4 + (size_t)(n ^ 1) * hw;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (t.off[k] >= 0) {
            const float4 v = b[t.off[k]];
            r0 = __fadd_rn(r0, __fmul_rn(v.x, t.w[k]));
            r1 = __fadd_rn(r1, __fmul_rn(v.y, t.w[k]));
            r2 = __fadd_rn(r2, __fmul_rn(v.z, t.w[k]));
        }
    float* o = out + idx * out_cs;
    o[0] = r0, o[1] = r1, o[2] = r2;
}

// workspace of the generic splat (tile ranges, max |f| per image) per (device, stream); grows, never shrinks
struct Splat4Ws {
    float4* ranges = nullptr;
    size_t n_ranges = 0;
    unsigned* smax = nullptr;
    size_t n_smax = 0;
};

int softsplat4_launch(const float* in, const float* flow, float* out, int N, int H, int W, hipStream_t s) {
    VFI_REQUIRE((long)H * W < (1L << 30), "softsplat: %d x %d pixels do not fit the 32-bit indices", H, W);
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "softsplat: device index %d out of range", dev);
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Splat4Ws> table;
    std::lock_guard<std::mutex> lock(mu);
    Splat4Ws& w = table[{dev, s}];
    const int tiles_x = cdiv(W, RT), tiles_y = cdiv(H, RT);
    const size_t nt = (size_t)N * tiles_x * tiles_y;
    if (w.n_ranges < nt) {
        // outgrown blocks are RETIRED, not freed: a captured HIP graph of the stream's owner may have their addresses baked in (r6)
        w.ranges = nullptr, w.n_ranges = 0;
        VFI_CHECK_HIP(hipMalloc((void**)&w.ranges, sizeof(float4) * nt));
        w.n_ranges = nt;
    }
    if (w.n_smax < (size_t)N) {
        w.smax = nullptr, w.n_smax = 0;
        VFI_CHECK_HIP(hipMalloc((void**)&w.smax, sizeof(unsigned) * (size_t)(N < 64 ? 64 : N)));
        w.n_smax = N < 64 ? 64 : N;
    }
    static std::atomic<int> attr_set[kMaxDevices];
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        VFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&softsplat4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSplat4Lds));
        attr_set[dev].store(1, std::memory_order_release);
    }
    VFI_CHECK_HIP(hipMemsetAsync(w.smax, 0, sizeof(unsigned) * (size_t)N, s));
    {
        TraceScope ts("splat_blockrange", s);
        hipLaunchKernelGGL(flow_tile_ranges_kernel, dim3((unsigned)nt), dim3(256), 0, s, (const float2*)flow, H, W, tiles_x, tiles_y, w.ranges, w.smax);
    }
    Splat4Args a;
    a.in = (const float4*)in, a.flow = (const float2*)flow, a.out = (float4*)out, a.brange = w.ranges, a.smax = w.smax;
    a.H = H, a.W = W, a.tiles_x = tiles_x, a.tiles_y = tiles_y;
    {
        TraceScope ts("softsplat_sum", s);
        hipLaunchKernelGGL(softsplat4_kernel, dim3((unsigned)nt), dim3(256), kSplat4Lds, s, a);
    }
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vfi

using namespace vfi;

extern "C" {

int vfi_m2m_image4(const float* d0_dev, int d0_cs, float* img4_dev, int H, int W, void* stream) {
    VFI_REQUIRE(d0_dev && img4_dev && d0_cs >= 5 && H > 0 && W > 0 && ((uintptr_t)img4_dev & 15) == 0, "vfi_m2m_image4: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("m2m_image4", s);
    hipLaunchKernelGGL(m2m_img4_kernel, dim3(nblk(2L * H * W)), dim3(256), 0, s, d0_dev, d0_cs, (float4*)img4_dev, 2L * H * W);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_m2m_warp_image4(const float* img4_dev, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int H, int W, void* stream) {
    VFI_REQUIRE(img4_dev && flow_dev && out_dev && flow_cs >= 2 && out_cs >= 3 && H > 1 && W > 1 && ((uintptr_t)img4_dev & 15) == 0,
                "vfi_m2m_warp_image4: bad arguments");
    float stepx, stepy, sclx, scly;
    m2m_warp_consts(H, W, stepx, stepy, sclx, scly);
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("warp_m2m", s);
    hipLaunchKernelGGL(m2m_warp_img4_kernel, dim3(nblk(2L * H * W)), dim3(256), 0, s, (const float4*)img4_dev, flow_dev, flow_cs, out_dev, out_cs, H, W,
                       stepx, stepy, sclx, scly);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_m2m_photo_tiles(const float* d0_dev, int d0_cs, const float* r_dev, int r_cs, float alpha, const float* img4_dev, float* tf_dev,
                        float* e_dev, float* tile_ranges_dev, float* smax_dev, int H, int W, void* stream) {
    VFI_REQUIRE(d0_dev && r_dev && img4_dev && tf_dev && e_dev && tile_ranges_dev && smax_dev && d0_cs >= 5 && r_cs >= 9 && H > 1 && W > 1,
                "vfi_m2m_photo_tiles: bad arguments");
    VFI_REQUIRE((((uintptr_t)img4_dev | (uintptr_t)tile_ranges_dev) & 15) == 0 && (((uintptr_t)tf_dev | (uintptr_t)d0_dev) & 7) == 0 && (d0_cs & 1) == 0,
                "vfi_m2m_photo_tiles: unaligned buffers (img4 / tile ranges 16 bytes, tf / d0 8 bytes, d0_cs even)");
    float stepx, stepy, sclx, scly;
    m2m_warp_consts(H, W, stepx, stepy, sclx, scly);
    hipStream_t s = (hipStream_t)stream;
    const int tiles_x = cdiv(W, RT), tiles_y = cdiv(H, RT);
    TraceScope ts("m2m_photo", s);
    VFI_CHECK_HIP(hipMemsetAsync(smax_dev, 0, 8 * sizeof(float), s));
    hipLaunchKernelGGL(m2m_photo_tiles_kernel, dim3(2 * tiles_x * tiles_y), dim3(256), 0, s, d0_dev, d0_cs, (const float4*)img4_dev, r_dev, r_cs,
                       alpha, tf_dev, e_dev, (float4*)tile_ranges_dev, (unsigned*)smax_dev, H, W, tiles_x, tiles_y, stepx, stepy, sclx, scly);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_m2m_render_fused(const float* img4_dev, const float* tf_dev, const float* e_dev, const float* tile_ranges_dev, const float* smax_dev,
                         const float* stats_dev, float t, float* out_dev, int Hp, int Wp, int H, int W, void* stream) {
    VFI_REQUIRE(img4_dev && tf_dev && e_dev && tile_ranges_dev && smax_dev && stats_dev && out_dev && Hp >= H && Wp >= W && H > 0 && W > 0,
                "vfi_m2m_render_fused: bad arguments");
    VFI_REQUIRE(t >= 0.f && t <= 1.f, "vfi_m2m_render_fused: timestep %g outside [0, 1] (the tile ranges scale with it)", (double)t);
    VFI_REQUIRE((long)Hp * Wp < (1L << 30), "vfi_m2m_render_fused: %d x %d pixels do not fit the 32-bit indices", Hp, Wp);
    hipStream_t s = (hipStream_t)stream;
    RenderArgs a;
    a.img4 = (const float4*)img4_dev;
    a.TF = tf_dev;
    a.E = e_dev;
    a.brange = (const float4*)tile_ranges_dev;
    a.smax = (const unsigned*)smax_dev;
    a.stats = stats_dev;
    a.out = out_dev;
    a.t = t;
    a.Hp = Hp, a.Wp = Wp, a.H = H, a.W = W;
    a.tiles_x = cdiv(Wp, RT), a.tiles_y = cdiv(Hp, RT);
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "vfi_m2m_render_fused: device index %d out of range", dev);
    static std::atomic<int> attr_set[kMaxDevices];
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        VFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&m2m_render_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRenderLds));
        attr_set[dev].store(1, std::memory_order_release);
    }
    TraceScope ts("m2m_render", s);
    hipLaunchKernelGGL(m2m_render_kernel, dim3((unsigned)(cdiv(a.tiles_x * a.tiles_y, 8) * 8)), dim3(256), kRenderLds, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
