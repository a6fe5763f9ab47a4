// Second-generation NHWC fp32 implicit-GEMM convolution for gfx950: BOTH operands are streamed
// HBM/L2 -> LDS by the LDS-DMA path (buffer_load_dwordx4 ... lds / global_load_lds_dwordx4), with a
// double-buffered K-chunk pipeline and one workgroup barrier per chunk.
//
// Why (profiles/r01_*_v1.txt): the first-generation kernel (conv_mfma.hip) keeps weights in VGPRs,
// loaded from L2 one K-step ahead; rocprofv3 shows the matrix pipe only 65 % busy on the block-3
// ResConv (SQ_WAIT_ANY = 27 % of wave cycles: exposed L2 latency in front of every 16-MFMA step).
// Here a K-step is  MT+NT ds_read_b128 + 4*MT*NT v_mfma_f32_32x32x2_f32  and nothing else; all global
// traffic for chunk k+1 is in flight (no VGPRs, no waits) while chunk k is multiplied.
//
//   * activations: halo'd input tile, CK channels per chunk, [pixel][CK] in LDS (lane-linear DMA
//     image); out-of-image halo pixels are zero-filled by the buffer descriptor's range check
//     (voffset >= num_records returns 0) — no branches, no separate memset;
//   * weights: host-packed [group][tap][Cin/8][Cout][8]; one (tap, 8-channel) slice of the block's
//     BN output channels is a contiguous run, copied verbatim: the LDS image IS the B fragment order;
//   * epilogue fused as in the first generation (+bias, *beta + residual, LeakyReLU, NHWC store).
#include "vfi_common.h"

#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace vfi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int STRIDE, int TAPS, int MT, int NT, int WM, int WN, int CK, bool GROUPED>
struct Conv2Geom {
    static constexpr int KW = TAPS == 9 ? 3 : (TAPS == 4 ? 2 : 1);
    static constexpr int SUBS = WM * MT;
    static constexpr int SUBX = SUBS >= 16 ? 4 : (SUBS >= 2 ? 2 : 1);   // 16 sub-tiles: 32 x 16 pixels (divides 480 x 272)
    static constexpr int SUBY = SUBS / SUBX;
    static constexpr int TWO = SUBX * 8, THO = SUBY * 4;
    // input tile: the rows / columns the taps reach.  3x3 (pad 1) and the grouped transposed conv (2x2 taps from origin -1 or 0)
    // need one halo pixel before and one after; 2x2 'same' (pads bottom / right) one after; 1x1 none
    static constexpr int PADLO = (TAPS == 9 || GROUPED) ? 1 : 0;
    static constexpr int SPAN = (TAPS == 9 || GROUPED) ? 3 : KW;
    static constexpr int TWI = STRIDE * (TWO - 1) + SPAN;
    static constexpr int THI = STRIDE * (THO - 1) + SPAN;
    static constexpr int NPIX = TWI * THI;
    static constexpr int Q = CK / 4;
    static constexpr int C8 = CK / 8;
    static constexpr int NA_INSTR = (NPIX * Q + 63) / 64;            // 1 KiB wave-wide DMA pieces
    static constexpr int A_FLOATS = NA_INSTR * 256;
    static constexpr int NGRP = GROUPED ? 4 : 1;
    static constexpr int BN = GROUPED ? 32 : WN * NT * 32;           // output channels per group per block
    static constexpr int NB_INSTR = NGRP * TAPS * C8 * (BN / 32);
    static constexpr int B_FLOATS = NB_INSTR * 256;
    static constexpr int BUF_FLOATS = A_FLOATS + B_FLOATS;
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
    static constexpr int NAW = (NA_INSTR + 3) / 4;                   // pieces per wave
    static constexpr int NBW = (NB_INSTR + 3) / 4;
};

// EXT = the layer uses the extended feature set (replicate padding, per-channel PReLU / sigmoid, post affine,
// interleaved transposed-conv store): compiled separately so the RIFE / FILM hot path carries none of its branches.
// MASKED (2x2 taps only; round 6): the layer is "nearest-neighbour up-sampling x2, then a 2x2 'same' convolution" (FILM's Fusion,
// film_arch.py:282-292) computed on the LOW-resolution input: output parity (py, px) of the up-sampled image reads low-resolution pixels
// (y + a, x + b), a <= py, b <= px, with the 2x2 weights summed over the taps that land on the same input pixel — 1 + 2 + 2 + 4 = 9 tap
// blocks instead of 16, no up-sampled tensor.  The four parities are 4 * Cout output channels; an N block belongs to one parity, walks only
// that parity's taps (a.tapmask[blockIdx.y]) and stores its pixels interleaved into the [2H, 2W] output.
template <int STRIDE, int TAPS, int MT, int NT, int WM, int WN, int CK, bool GROUPED, bool EXT, bool MASKED = false>
__global__ __launch_bounds__(256) void conv_mfma2_kernel(const ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-resource / LDS-DMA builtins do not exist in the host pass
    using G = Conv2Geom<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED>;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(!GROUPED || (WN == 4 && NT == 1 && TAPS == 4), "grouped: one 2x2 tap group per wave column");
    constexpr int KW = G::KW, SUBX = G::SUBX, TWO = G::TWO, THO = G::THO, TWI = G::TWI;
    constexpr int Q = G::Q, C8 = G::C8, BN = G::BN;

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    const int g = GROUPED ? wn : 0;
    const int cob = blockIdx.y * BN;                       // first output channel of this block (per group)
    const int co0 = GROUPED ? cob : cob + wn * NT * 32;    // first output channel of this wave
    const int cin8 = a.Cin_p >> 3;
    const int tby = GROUPED ? (g >> 1) - 1 : a.tap_y0;
    const int tbx = GROUPED ? (g & 1) - 1 : a.tap_x0;

    // ---- persistent tile loop: this workgroup owns M tiles blockIdx.x, +gridDim.x, ...
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int T = a.N * tiles_per_img;
    const int pstr = a.in_plane ? 4 : a.in_cs;                 // floats between pixels
    const int qstr = a.in_plane ? a.in_plane : 4;              // floats between 4-channel groups
    const int cadv = a.in_plane ? (CK / 4) * a.in_plane : CK;  // floats between K chunks
    const int img_floats = a.in_plane ? (a.in_cs / 4) * a.in_plane : a.Hin * a.Win * a.in_cs;
    auto decode = [&](int tile, int& n, int& Y0, int& X0) {
        n = tile / tiles_per_img;
        const int trem = tile - n * tiles_per_img;
        const int ty = trem / a.tiles_x;
        Y0 = ty * THO;
        X0 = (trem - ty * a.tiles_x) * TWO;
    };
    // activation DMA: voffset per 16-byte piece inside the image; out-of-image halo -> out of range -> zero fill
    auto make_avoff = [&](int Y0, int X0, int(&avoff)[G::NAW]) {
        const int iy0 = STRIDE * Y0 - G::PADLO, ix0 = STRIDE * X0 - G::PADLO;
#pragma unroll
        for (int i = 0; i < G::NAW; ++i) {
            const int j = wave + 4 * i;
            const int idx = j * 64 + lane;
            const int pix = idx / Q, q = idx - pix * Q;
            const int py = pix / TWI, px = pix - py * TWI;
            const int iy = iy0 + py, ix = ix0 + px;
            if (EXT && a.pad_replicate) {  // edge clamp instead of the descriptor's zero fill
                const int cy = min(max(iy, 0), a.Hin - 1), cx = min(max(ix, 0), a.Win - 1);
                avoff[i] = pix < G::NPIX ? ((cy * a.Win + cx) * pstr + q * qstr) * 4 : (int)0x80000000;
            } else {
                const bool ok = pix < G::NPIX && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                avoff[i] = ok ? ((iy * a.Win + ix) * pstr + q * qstr) * 4 : (int)0x80000000;
            }
        }
    };
    auto make_rsrc = [&](int n) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * img_floats), 0, img_floats * 4, 0x00020000);
    };
    // weight DMA: per-lane source pointers of this wave's pieces (tile- and chunk-invariant part)
    const unsigned tmask = MASKED ? a.tapmask[blockIdx.y] : 0xffffu;
    const int tapstride = cin8 * a.Cout_p * 8;
    const int c8stride = a.Cout_p * 8;
    const float* wsrc[G::NBW];
    bool wlive[G::NBW];
#pragma unroll
    for (int i = 0; i < G::NBW; ++i) {
        const int j = wave + 4 * i;  // piece index in LDS order [group][tap][c8][BN/32]
        const int sub = j % (BN / 32);
        const int c8 = (j / (BN / 32)) % C8;
        const int t = (j / (BN / 32) / C8) % TAPS;
        const int gg = j / (BN / 32) / C8 / TAPS;
        wsrc[i] = a.w + (size_t)(gg * TAPS + t) * tapstride + c8 * c8stride + (cob + sub * 32) * 8 + lane * 4;
        wlive[i] = !MASKED || ((tmask >> t) & 1u);      // (wave-uniform)
    }
    auto issue = [&](const __amdgpu_buffer_rsrc_t& rsrc, const int(&avoff)[G::NAW], int chunk, int buf) {
        float* abuf = smem + buf * G::BUF_FLOATS;
        float* bbuf = abuf + G::A_FLOATS;
        const int cbyte = chunk * cadv * 4;
#pragma unroll
        for (int i = 0; i < G::NAW; ++i) {
            const int j = wave + 4 * i;
            if (G::NA_INSTR % 4 == 0 || j < G::NA_INSTR)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(abuf + j * 256), 16, avoff[i] + cbyte, 0, 0, 0);
        }
        const int wadv = chunk * C8 * c8stride;
#pragma unroll
        for (int i = 0; i < G::NBW; ++i) {
            const int j = wave + 4 * i;
            if ((G::NB_INSTR % 4 == 0 || j < G::NB_INSTR) && wlive[i])
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(wsrc[i] + wadv), (lds_ptr_t)(bbuf + j * 256), 16, 0, 0);
        }
    };

    // ---- fragment offsets (floats) inside a buffer
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int s = wm * MT + mt;
        const int sx = s % SUBX, sy = s / SUBX;
        const int oy = sy * 4 + (l31 >> 3), ox = sx * 8 + (l31 & 7);
        abase[mt] = ((STRIDE * oy + G::PADLO + tby) * TWI + STRIDE * ox + G::PADLO + tbx) * CK + half * 4;
    }
    // B image: [group][tap][c8][BN][8]; this wave's channels start at (co0 - cob)
    const int bbase = G::A_FLOATS + (g * TAPS * C8 * BN + (co0 - cob) + l31) * 8 + half * 4;

    // per-lane epilogue constants (the lane's output channel is fixed per N sub-tile)
    float bs[NT], bt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + nt * 32 + l31;
        bs[nt] = a.bias[g * a.Cout_p + co];
        bt[nt] = a.beta ? a.beta[co] : 1.f;
    }

    // split-K (small images with many input channels, see launch2_e): blockIdx.z owns the K chunks [k0, k0 + nchunks) and
    // writes its partial sums to its own slice of a workspace; conv2_split_reduce_kernel adds the slices and applies the epilogue
    int nchunks = a.Cin_p / CK, k0 = 0;
    float* const outp = a.out + (size_t)blockIdx.z * a.split_stride;
    if (a.ksplit > 1) {
        const int per = (nchunks + a.ksplit - 1) / a.ksplit;
        k0 = blockIdx.z * per;
        nchunks = nchunks - k0 < per ? nchunks - k0 : per;
    }

    int tile = blockIdx.x;
    if (tile >= T) return;
    int n, Y0, X0, avoff[G::NAW];
    decode(tile, n, Y0, X0);
    make_avoff(Y0, X0, avoff);
    __amdgpu_buffer_rsrc_t rsrc = make_rsrc(n);
    issue(rsrc, avoff, k0, 0);
    __syncthreads();  // LDS-DMA counts on vmcnt: the barrier's release drains it
    int buf = 0;
    for (;;) {
        const int ntile = tile + gridDim.x;
        const bool has_next = ntile < T;
        int nn = 0, nY0 = 0, nX0 = 0, navoff[G::NAW];
        __amdgpu_buffer_rsrc_t nrsrc = rsrc;

        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

        for (int k = 0; k < nchunks; ++k) {
            // keep the DMA queue one K-chunk ahead — across the tile boundary too, so the matrix pipe
            // never waits for a tile's first chunk and the previous tile's stores drain underneath
            if (k + 1 < nchunks) {
                issue(rsrc, avoff, k0 + k + 1, buf ^ 1);
            } else if (has_next) {
                decode(ntile, nn, nY0, nX0);
                make_avoff(nY0, nX0, navoff);
                nrsrc = make_rsrc(nn);
                issue(nrsrc, navoff, k0, buf ^ 1);
            }
            const float* sb = smem + buf * G::BUF_FLOATS;
            // K-steps of this chunk, software-pipelined: the fragments of step s+1 are requested from LDS
            // before the MFMAs of step s issue (two register sets), so one wave alone keeps the matrix pipe fed.
            constexpr int NS = TAPS * C8;
            f32x4 av[2][MT], bv[2][NT];
            auto frag = [&](int st, f32x4(&fa)[MT], f32x4(&fb)[NT]) {
                const int t = st / C8, c8 = st % C8;
                const int toff = ((t / KW) * TWI + (t % KW)) * CK + c8 * 8;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) fa[mt] = *(const f32x4*)&sb[abase[mt] + toff];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    fb[nt] = *(const f32x4*)&sb[bbase + ((t * C8 + c8) * BN + nt * 32) * 8];
            };
            if constexpr (MASKED) {
                // the parity's taps only: 1, 2 or 4 of the 4 (C8 == 1: one K-step per tap); same pipelining, step list from the mask
                static_assert(!MASKED || TAPS == 4, "masked form: 2x2 taps");
                // (wave-uniform scalars, no indexed writes: lowest set bits of the mask in turn)
                const unsigned m0 = tmask & 15u, m1 = m0 & (m0 - 1u), m2 = m1 & (m1 - 1u), m3 = m2 & (m2 - 1u);
                const int tl[4] = {m0 ? __builtin_ctz(m0) : 0, m1 ? __builtin_ctz(m1) : 0, m2 ? __builtin_ctz(m2) : 0, m3 ? __builtin_ctz(m3) : 0};
                const int ntl = __builtin_popcount(m0);
                auto run = [&](auto NTL) {
                    constexpr int n_ = decltype(NTL)::value * C8;      // K-steps of this chunk: the parity's taps x the chunk's 8-channel groups
                    auto step_of = [&](int i) { return tl[i / C8] * C8 + i % C8; };
                    frag(step_of(0), av[0], bv[0]);
#pragma unroll
                    for (int i = 0; i < n_; ++i) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i & 1][mt][j], bv[i & 1][nt][j], acc[mt][nt], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            if (j == 0 && i + 1 < n_) {
                                frag(step_of(i + 1), av[(i + 1) & 1], bv[(i + 1) & 1]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                };
                if (ntl == 1) run(std::integral_constant<int, 1>{});
                else if (ntl == 2) run(std::integral_constant<int, 2>{});
                else run(std::integral_constant<int, 4>{});
            } else {
            frag(0, av[0], bv[0]);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                // Issue order matters twice.  (1) Consecutive MFMAs must rotate through ALL MT*NT accumulators: left alone, hipcc
                // groups them by A operand (acc0, acc1, acc0, acc1, ...), halving the distance between dependent MFMAs.
                // (2) The fragment reads of step s+1 are issued after the first accumulator round of step s: early enough to
                // hide their latency under the remaining 3/4 of the step, late enough that the s_waitcnt lgkmcnt(0) hipcc puts
                // in front of the step's first MFMA (it does not count the in-order LDS returns) only sees reads that had a
                // whole step to complete.  (Pure-MFMA microbenchmark: 154 TFLOP/s; this loop before pinning: 122.)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st & 1][mt][j], bv[st & 1][nt][j],
                                                                               acc[mt][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j == 0 && st + 1 < NS) {
                        frag(st + 1, av[(st + 1) & 1], bv[(st + 1) & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            }
            __syncthreads();  // chunk consumed by every wave; the next chunk has landed
            buf ^= 1;
        }

        // ---- epilogue (stores are fire-and-forget; the next tile's MFMAs start right behind them)
        // Fast path, decided once per tile with uniform branches: plain NHWC store of an interior tile, activation none /
        // LeakyReLU with a slope in [0,1] (then lrelu(v) == max(v, v*slope)), residual folded.  The generic path below
        // re-tests act / res / beta / bounds for each of the 64 values of a lane — ~4000 instructions and ~700 scalar branches
        // per tile, during which this wave feeds no MFMAs.
        if constexpr (MASKED) {
            // parity-interleaved store: N block -> parity g = first channel / par_cout; low-resolution pixel (oy, ox) -> (2 oy + gy, 2 ox + gx)
            const int gpar = cob / a.par_cout, gy = gpar >> 1, gx = gpar & 1;
            const int xstr = 2 * a.out_cs, ystr = 4 * a.Wout * a.out_cs;
            const bool interior = Y0 + THO <= a.Hout && X0 + TWO <= a.Wout;
            const float uslope = a.act == 1 ? a.slope : 1.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = co0 + nt * 32 + l31;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int s = wm * MT + mt;
                    const int sx = s % SUBX, sy = s / SUBX;
                    const int oy0 = Y0 + sy * 4, ox0 = X0 + sx * 8 + 4 * half;
                    float* ob = outp + ((size_t)(n * 2 * a.Hout + 2 * oy0 + gy) * (2 * a.Wout) + 2 * ox0 + gx) * a.out_cs + (co - gpar * a.par_cout);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[mt][nt][r] + bs[nt];
                        v = v > 0.f ? v : v * uslope;
                        if (interior || (oy0 + (r >> 2) < a.Hout && ox0 + (r & 3) < a.Wout)) ob[(r >> 2) * ystr + (r & 3) * xstr] = v;
                    }
                }
            }
        } else {
        const bool nhwc = a.out_mode == 0 && !GROUPED;
        const bool inter = EXT && GROUPED && a.out_mode == 2;       // transposed conv, parity groups interleaved into NHWC
        const bool fast = a.res == nullptr && (nhwc || inter) && (a.act == 0 || a.act == 1 || (EXT && (a.act == 3 || a.act == 5))) &&
                          !(EXT && a.post_scale != 0.f);
        if (fast) {
            // element (r) of a sub-tile -> output pixel: NHWC (oy, ox); interleaved (2*oy + gy, 2*ox + gx) of a [2H, 2W] image
            const int m = inter ? 2 : 1;
            const int xstr = m * a.out_cs, ystr = m * m * a.Wout * a.out_cs;
            const bool interior = Y0 + THO <= a.Hout && X0 + TWO <= a.Wout;
            const bool unif01 = a.act == 0 || (a.act == 1 && a.slope >= 0.f && a.slope <= 1.f);
            const float uslope = a.act == 1 ? a.slope : 1.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = co0 + nt * 32 + l31;
                if (co < a.Cout) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int s = wm * MT + mt;
                        const int sx = s % SUBX, sy = s / SUBX;
                        const int oy0 = Y0 + sy * 4, ox0 = X0 + sx * 8 + 4 * half;
                        float* ob = outp + ((size_t)(n * m * a.Hout + m * oy0 + (inter ? (g >> 1) : 0)) * (m * a.Wout) + m * ox0 +
                                             (inter ? (g & 1) : 0)) * a.out_cs + co;
                        if (unif01 && interior) {   // the common case: uniform slope in [0,1] -> lrelu(v) == max(v, v*slope)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = (acc[mt][nt][r] + bs[nt]) * bt[nt];
                                ob[(r >> 2) * ystr + (r & 3) * xstr] = fmaxf(v, v * uslope);
                            }
                        } else {                    // image border and / or per-channel (or out-of-range) slopes
                            const bool gelu = EXT && a.act == 5;
                            const float sl = a.act == 0 || gelu ? 1.0f : (a.act == 1 ? a.slope : a.prelu[co]);
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                float v = (acc[mt][nt][r] + bs[nt]) * bt[nt];
                                v = v > 0.f ? v : v * sl;
                                if (gelu) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));   // nn.GELU(), erf form
                                if (interior || (oy0 + (r >> 2) < a.Hout && ox0 + (r & 3) < a.Wout)) ob[(r >> 2) * ystr + (r & 3) * xstr] = v;
                            }
                        }
                    }
                }
            }
        } else if (!EXT && GROUPED && a.out_mode == 1 && a.act == 0 && a.beta == nullptr && a.res == nullptr) {
            // transposed conv + PixelShuffle(2) into the planar4 block output (see conv_mfma.hip)
            const int Ws = 4 * a.Wout, Hs = 4 * a.Hout;
            const int planes = a.out_planes ? a.out_planes : 2;
            const bool interior = Y0 + THO <= a.Hout && X0 + TWO <= a.Wout;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = co0 + nt * 32 + l31;
                if (co < a.Cout) {
                    const int c = co >> 2;
                    float* lane_base = outp + ((size_t)(n * planes + (c >> 2)) * Hs * Ws + (size_t)(2 * (g >> 1) + ((co >> 1) & 1)) * Ws +
                                                2 * (g & 1) + (co & 1)) * 4 + (c & 3);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const int s = wm * MT + mt;
                        const int sx = s % SUBX, sy = s / SUBX;
                        const int oy0 = Y0 + sy * 4, ox0 = X0 + sx * 8 + 4 * half;
                        float* ob = lane_base + ((size_t)(4 * oy0) * Ws + 4 * ox0) * 4;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (interior || (oy0 + (r >> 2) < a.Hout && ox0 + (r & 3) < a.Wout))
                                ob[((r >> 2) * 4 * Ws + (r & 3) * 4) * 4] = acc[mt][nt][r] + bs[nt];
                    }
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = co0 + nt * 32 + l31;
                const bool cok = co < a.Cout;
                const int coc = cok ? co : a.Cout - 1;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int s = wm * MT + mt;
                    const int sx = s % SUBX, sy = s / SUBX;
                    // explicit residual (test / non-folded path): 16 loads in flight, clamped addresses, one wait
                    float rv[16];
                    if (a.res) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            int oy = Y0 + sy * 4 + (r >> 2), ox = X0 + sx * 8 + (r & 3) + 4 * half;
                            oy = oy < a.Hout ? oy : a.Hout - 1;
                            ox = ox < a.Wout ? ox : a.Wout - 1;
                            rv[r] = a.res[((size_t)(n * a.Hout + oy) * a.Wout + ox) * a.res_cs + coc];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int oy = Y0 + sy * 4 + (r >> 2);
                        const int ox = X0 + sx * 8 + (r & 3) + 4 * half;
                        float v = acc[mt][nt][r] + bs[nt];
                        if (a.beta) v *= bt[nt];
                        if (a.res) v += rv[r];
                        if (a.act == 1) v = v > 0.f ? v : v * a.slope;
                        else if (a.act == 2) v = fminf(fmaxf(v, 0.f), 1.f);
                        else if (EXT && a.act == 3) v = v > 0.f ? v : v * a.prelu[coc];
                        else if (EXT && a.act == 4) v = 1.0f / (1.0f + expf(-v));
                        else if (EXT && a.act == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                        if (EXT && a.post_scale != 0.f) v = v * a.post_scale + a.post_shift;
                        if (cok && oy < a.Hout && ox < a.Wout) {
                            if (GROUPED && a.out_mode == 1) {
                                const int Ws = 4 * a.Wout, Hs = 4 * a.Hout, c = co >> 2;
                                const int Yt = 4 * oy + 2 * (g >> 1) + ((co >> 1) & 1), Xt = 4 * ox + 2 * (g & 1) + (co & 1);
                                outp[((size_t)(n * (a.out_planes ? a.out_planes : 2) + (c >> 2)) * Hs * Ws + (size_t)Yt * Ws + Xt) * 4 + (c & 3)] = v;
                            } else if (EXT && GROUPED && a.out_mode == 2) {
                                const size_t q2 = (size_t)(n * 2 * a.Hout + 2 * oy + (g >> 1)) * (2 * a.Wout) + 2 * ox + (g & 1);
                                outp[q2 * a.out_cs + co] = v;
                            } else {
                                outp[((size_t)(n * a.Hout + oy) * a.Wout + ox) * a.out_cs + g * a.Cout_p + co] = v;
                            }
                        }
                    }
                }
            }
        }
        }
        if (!has_next) break;
        tile = ntile;
        n = nn;
        Y0 = nY0;
        X0 = nX0;
        rsrc = nrsrc;
#pragma unroll
        for (int i = 0; i < G::NAW; ++i) avoff[i] = navoff[i];
    }
#endif
}

// ---- split-K support -----------------------------------------------------------------------------------------------------
// out = epilogue(sum_z ws[z] + bias): the epilogue of the generic path of conv_mfma2_kernel, element by element
__global__ __launch_bounds__(256) void conv2_split_reduce_kernel(const ConvArgs a, const float* __restrict__ ws, int ks, long slice) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;     // over [pixels][Cout]
    const long total = (long)a.N * a.Hout * a.Wout * a.Cout;
    if (idx >= total) return;
    const long p = idx / a.Cout;
    const int co = (int)(idx - p * a.Cout);
    float v = 0.f;
    for (int z = 0; z < ks; ++z) v += ws[z * slice + p * a.Cout_p + co];
    v += a.bias[co];
    if (a.beta) v *= a.beta[co];
    if (a.res) v += a.res[p * a.res_cs + co];
    if (a.act == 1) v = v > 0.f ? v : v * a.slope;
    else if (a.act == 2) v = fminf(fmaxf(v, 0.f), 1.f);
    else if (a.act == 3) v = v > 0.f ? v : v * a.prelu[co];
    else if (a.act == 4) v = 1.0f / (1.0f + expf(-v));
    else if (a.act == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    if (a.post_scale != 0.f) v = v * a.post_scale + a.post_shift;
    a.out[p * a.out_cs + co] = v;
}

static bool split_enabled() { return option(kOptSplitK) != 0; }      // A/B option splitk = 0 disables split-K

// partial-sum workspace and a zero bias vector, per (device, stream): grow-only, kept for the life of the process
static int split_workspace(int dev, hipStream_t s, size_t floats, int cout_p, float** ws, float** zeros) {
    struct Ws {
        float* p = nullptr;
        size_t n = 0;
        float* z = nullptr;
        int zn = 0;
    };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Ws> table;
    std::lock_guard<std::mutex> lock(mu);
    Ws& w = table[{dev, s}];
    if (w.n < floats) {
        // the outgrown block is RETIRED, not freed: a captured HIP graph of the stream's owner may have its address baked in (r6), and
        // hipFree would drain the device under the other pair lanes (sizes grow a handful of times in a process)
        w.p = nullptr, w.n = 0;
        VFI_CHECK_HIP(hipMalloc((void**)&w.p, floats * sizeof(float)));
        w.n = floats;
    }
    if (w.zn < cout_p) {
        w.z = nullptr, w.zn = 0;      // (retired likewise)
        const int n = cout_p < 4096 ? 4096 : cout_p;
        VFI_CHECK_HIP(hipMalloc((void**)&w.z, n * sizeof(float)));
        // on the launch's own stream: a NULL-stream memset is not ordered against a non-blocking stream, and under load (other pair
        // lanes keeping the device busy) it landed AFTER the first split launch had read the vector as its bias (r6: IFRNet, lane 1's
        // first pair off by 5e-2)
        VFI_CHECK_HIP(hipMemsetAsync(w.z, 0, n * sizeof(float), s));
        w.zn = n;
    }
    *ws = w.p, *zeros = w.z;
    return 0;
}

static int split_reduce(const ConvArgs& a, const float* ws, int ks, size_t slice, hipStream_t s) {
    TraceScope ts("conv_split_reduce", s);
    const long total = (long)a.N * a.Hout * a.Wout * a.Cout;
    hipLaunchKernelGGL(conv2_split_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, ws, ks, (long)slice);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int STRIDE, int TAPS, int MT, int NT, int WM, int WN, int CK, bool GROUPED, bool EXT, bool MASKED = false>
static int launch2_e(ConvArgs a, hipStream_t s, const char* name) {
    using G = Conv2Geom<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED>;
    a.ksplit = 0, a.split_stride = 0;      // launcher-owned fields (callers do not set them)
    a.tiles_x = cdiv(a.Wout, G::TWO);
    a.tiles_y = cdiv(a.Hout, G::THO);
    VFI_REQUIRE(a.Cin_p % CK == 0, "conv2 %s: Cin_p=%d not a multiple of the K chunk %d", name, a.Cin_p, CK);
    VFI_REQUIRE(a.Cout_p % G::BN == 0, "conv2 %s: Cout_p=%d not a multiple of the N tile %d", name, a.Cout_p, G::BN);
    VFI_REQUIRE((long)a.Hin * a.Win * a.in_cs * 4 < 0x7fffffffL, "conv2 %s: image larger than 2 GiB", name);
    VFI_REQUIRE(!a.in_plane || a.in_plane >= a.Hin * a.Win * 4, "conv2 %s: bad plane stride", name);
    // resident workgroups per CU of this instantiation and the CU count, PER DEVICE: one process may drive several
    // devices from several threads (multidev.py), and hipFuncSetAttribute applies to the current device only
    static std::atomic<int> occ_of[kMaxDevices];
    static std::atomic<int> cus_of[kMaxDevices];
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "conv2 %s: device index %d out of range", name, dev);
    int occ = occ_of[dev].load(std::memory_order_acquire);
    if (!occ) {
        VFI_CHECK_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void*>(&conv_mfma2_kernel<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED, EXT, MASKED>),
            hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        int o = 0;
        VFI_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &o, reinterpret_cast<const void*>(&conv_mfma2_kernel<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED, EXT, MASKED>), 256,
            G::LDS_BYTES));
        hipDeviceProp_t p;
        VFI_CHECK_HIP(hipGetDeviceProperties(&p, dev));
        cus_of[dev].store(p.multiProcessorCount, std::memory_order_relaxed);
        occ = o < 1 ? 1 : o;
        occ_of[dev].store(occ, std::memory_order_release);   // (two threads racing here compute the same values)
    }
    const int cus = cus_of[dev].load(std::memory_order_relaxed);
    const int T = a.N * a.tiles_x * a.tiles_y;
    const int ny = a.Cout_p / G::BN;
    // ---- split-K: a coarse pyramid level with many input channels (FILM's 1920 -> 256 at 16 x 30: 32 workgroups, each walking
    // 240 K chunks: 7.9 TFLOP/s) is cut along K into `ks` launches-in-one (gridDim.z), partial sums to a workspace, then one
    // reduce + epilogue kernel; fixed summation order, so still deterministic.  Chosen so that the grid reaches ~3 workgroups
    // per slot while every split keeps >= 8 chunks (the DMA pipeline's prologue / epilogue amortised).
    int ks = 1;
    if (!GROUPED && !MASKED && a.out_mode == 0 && a.split_ok && split_enabled()) {
        const int nchunks = a.Cin_p / CK;
        const long wgs = (long)T * ny, want = 3L * cus * occ;
        // ... and a K loop long enough that a fraction of it outweighs the reduce launch (~20 us): 32 chunks of a 3x3 layer
        // (IFUNet's 128-channel layers at 16 chunks lost 0.7 ms per frame to their reduces)
        // — or so few workgroups that the layer leaves 7/8 of the chip idle (FILM's 128-channel layers at 16 x 30 and 33 x 60)
        const bool long_k = TAPS * nchunks >= 288, tiny = wgs * 8 <= (long)cus * occ && nchunks >= 16;
        if (wgs < (long)cus * occ && (long_k || tiny)) {      // less than one resident wave of workgroups
            ks = (int)((want + wgs - 1) / wgs);
            ks = ks > nchunks / 8 ? nchunks / 8 : ks;
            ks = ks > 16 ? 16 : ks;
            // the partial sums cost 8 ks bytes of HBM traffic per output value against 2 TAPS Cin flops: keep that below ~1/4 of
            // the layer's time (measured on IFUNet's 256 -> 256 layers at 68 x 120: ks = 6 gave back all it gained)
            const int ks_traffic = TAPS * a.Cin_p / 576;
            ks = ks > ks_traffic ? ks_traffic : ks;
            ks = ks < 1 ? 1 : ks;
            const size_t slice = (size_t)a.N * a.Hout * a.Wout * a.Cout_p;
            while (ks > 1 && slice * ks * sizeof(float) > (256u << 20)) --ks;
            const int per = (nchunks + ks - 1) / ks;
            ks = (nchunks + per - 1) / per;          // no empty split
        }
    }
    if (ks > 1) {
        const size_t slice = (size_t)a.N * a.Hout * a.Wout * a.Cout_p;
        float *ws = nullptr, *zeros = nullptr;
        if (int rc = split_workspace(dev, s, slice * ks, a.Cout_p, &ws, &zeros)) return rc;
        ConvArgs p = a;                              // the partial-sum launch: raw accumulators, dense [N,H,W,Cout_p]
        p.out = ws, p.out_cs = a.Cout_p, p.bias = zeros, p.beta = nullptr, p.res = nullptr, p.res_cs = 0;
        p.act = 0, p.post_scale = 0.f, p.post_shift = 0.f, p.Cout = a.Cout_p;
        p.ksplit = ks, p.split_stride = (long)slice;
        {
            TraceScope ts(name, s);
            // (the same instantiation as the unsplit launch: its dynamic-LDS attribute is the one set above; pad_replicate implies EXT)
            hipLaunchKernelGGL((conv_mfma2_kernel<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED, EXT, MASKED>), dim3(T, ny, ks), dim3(256), G::LDS_BYTES, s, p);
            VFI_CHECK_HIP(hipGetLastError());
        }
        return split_reduce(a, ws, ks, slice, s);
    }
    // Persistent (one resident wave of workgroups, each walks its share of tiles, DMA pipelined across tiles)
    // when every workgroup gets several tiles; with only a few tiles per slot a static split leaves some
    // CUs a whole tile behind, so then launch one workgroup per tile and let the dispatcher balance.
    int gx = (launch_cus(cus) * occ) / ny;
    if (gx < 1) gx = 1;
    if (gx > T || (long)T * ny < 4L * cus * occ) gx = T;
    if (MASKED) gx = T;      // N blocks carry 1, 2, 2 or 4 taps: one workgroup per tile, the dispatcher balances them
    dim3 grid(gx, ny);
    TraceScope ts(name, s);
    hipLaunchKernelGGL((conv_mfma2_kernel<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED, EXT, MASKED>), grid, dim3(256), G::LDS_BYTES, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

template <int STRIDE, int TAPS, int MT, int NT, int WM, int WN, int CK>
static int launch2_masked(const ConvArgs& a, hipStream_t s, const char* name) {
    VFI_REQUIRE(a.tapmask && a.par_cout > 0 && a.par_cout % (WN * NT * 32) == 0 && a.Cout_p == 4 * a.par_cout && a.act <= 1 && !a.res && !a.beta && !a.in_plane,
                "conv2 %s: masked (up-sample x2 + 2x2) form needs 4 parity groups of a multiple of %d channels, act none / LeakyReLU", name, WN * NT * 32);
    return launch2_e<STRIDE, TAPS, MT, NT, WM, WN, CK, false, false, true>(a, s, name);
}

template <int STRIDE, int TAPS, int MT, int NT, int WM, int WN, int CK, bool GROUPED>
static int launch2_t(const ConvArgs& a, hipStream_t s, const char* name) {
    const bool ext = a.pad_replicate || a.act >= 3 || a.post_scale != 0.f || a.out_mode == 2;
    return ext ? launch2_e<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED, true>(a, s, name)
               : launch2_e<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED, false>(a, s, name);
}

// second-generation variants, numbered from kConv2Base in the common variant space
static const ConvVariant kVariants2[] = {
    // name            stride taps mt nt wm wn ck grouped
    {"d1_m2n2", 1, 9, 2, 2, 4, 1, 8, 0},      // 32: 16x16 px x 64 ch
    {"d1_m2n3", 1, 9, 2, 3, 4, 1, 8, 0},      // 33: 16x16 px x 96 ch
    {"d1_m1n2", 1, 9, 1, 2, 4, 1, 8, 0},      // 34: 16x8 px x 64 ch
    {"d1_m1n3", 1, 9, 1, 3, 4, 1, 8, 0},      // 35
    {"d1_m2n2w22", 1, 9, 2, 2, 2, 2, 8, 0},   // 36: 16x8 px x 128 ch
    {"d1_m4n2w22", 1, 9, 4, 2, 2, 2, 8, 0},   // 37: 16x16 px x 128 ch
    {"d1_m2n2k16", 1, 9, 2, 2, 4, 1, 16, 0},  // 38: as 32 with 16-channel chunks
    {"d2_m1n2", 2, 9, 1, 2, 4, 1, 8, 0},      // 39: stride 2, 16x8 px x 64 ch
    {"d2_m2n2", 2, 9, 2, 2, 4, 1, 8, 0},      // 40
    {"d2_m1n3", 2, 9, 1, 3, 4, 1, 8, 0},      // 41
    {"d2_m2n1", 2, 9, 2, 1, 4, 1, 8, 0},      // 42: stride 2, 16x16 px x 32 ch
    {"dg_m4", 1, 4, 4, 1, 1, 4, 8, 1},        // 43: grouped 16x8 px
    {"dg_m8", 1, 4, 8, 1, 1, 4, 8, 1},        // 44: grouped 16x16 px
    {"d1_m2n1", 1, 9, 2, 1, 4, 1, 8, 0},      // 45: 3x3, 16x16 px x 32 ch
    {"d1t4_m2n2", 1, 4, 2, 2, 4, 1, 8, 0},    // 46: 2x2 taps (FILM 'same' conv, pad bottom/right)
    {"d1t4_m1n2", 1, 4, 1, 2, 4, 1, 8, 0},    // 47
    {"d1t1_m2n2", 1, 1, 2, 2, 4, 1, 8, 0},    // 48: 1x1
    {"d1t1_m1n2", 1, 1, 1, 2, 4, 1, 8, 0},    // 49
    {"d1t1_m2n1", 1, 1, 2, 1, 4, 1, 8, 0},    // 50: 1x1, 32-channel N tile
    {"d1t4_m2n1", 1, 4, 2, 1, 4, 1, 8, 0},    // 51
    {"d2t4_m2n1", 2, 4, 2, 1, 4, 1, 8, 0},    // 52: 2x2 stride 2 (M2M 'sconv(2)'), 32-channel N tile
    {"d2t4_m1n2", 2, 4, 1, 2, 4, 1, 8, 0},    // 53
    {"d1_m4n2", 1, 9, 4, 2, 4, 1, 8, 0},      // 54: 32x16 px x 64 ch (8 accumulators per wave)
    {"d1t1_m1n2k32", 1, 1, 1, 2, 4, 1, 32, 0},      // 55: 1x1 with 32-channel K chunks (4 K-steps per barrier instead of 1), 16x8 px x 64 ch
    {"d1t1_m2n2w22k32", 1, 1, 2, 2, 2, 2, 32, 0},   // 56: ... 16x8 px x 128 ch (wide inputs)
    {"d1t4_m2n2_up2", 1, 4, 2, 2, 4, 1, 8, 0},      // 57: nearest x2 + 2x2 'same' on the low-resolution input, per-parity tap masks (FILM Fusion, r6)
    {"d1t4_m1n2_up2", 1, 4, 1, 2, 4, 1, 8, 0},      // 58
    {"d1t4_m2n2k16_up2", 1, 4, 2, 2, 4, 1, 16, 0},  // 59: ... with 16-channel chunks (2 .. 8 K-steps per barrier instead of 1 .. 4)
    {"d1t4_m1n2k16_up2", 1, 4, 1, 2, 4, 1, 16, 0},  // 60
    {"d1_m1n1", 1, 9, 1, 1, 4, 1, 8, 0},            // 61: 3x3, 16x8 px x 32 ch — coarse pyramid levels (r6): twice the workgroups, half the serial K loop
};
int conv2_num_variants() { return (int)(sizeof(kVariants2) / sizeof(kVariants2[0])); }
const ConvVariant& conv2_variant(int i) { return kVariants2[i]; }

int conv2_launch(const ConvArgs& a, int idx, hipStream_t s, const char* nm) {
    switch (idx) {
        case 0: return launch2_t<1, 9, 2, 2, 4, 1, 8, false>(a, s, nm);
        case 1: return launch2_t<1, 9, 2, 3, 4, 1, 8, false>(a, s, nm);
        case 2: return launch2_t<1, 9, 1, 2, 4, 1, 8, false>(a, s, nm);
        case 3: return launch2_t<1, 9, 1, 3, 4, 1, 8, false>(a, s, nm);
        case 4: return launch2_t<1, 9, 2, 2, 2, 2, 8, false>(a, s, nm);
        case 5: return launch2_t<1, 9, 4, 2, 2, 2, 8, false>(a, s, nm);
        case 6: return launch2_t<1, 9, 2, 2, 4, 1, 16, false>(a, s, nm);
        case 7: return launch2_t<2, 9, 1, 2, 4, 1, 8, false>(a, s, nm);
        case 8: return launch2_t<2, 9, 2, 2, 4, 1, 8, false>(a, s, nm);
        case 9: return launch2_t<2, 9, 1, 3, 4, 1, 8, false>(a, s, nm);
        case 10: return launch2_t<2, 9, 2, 1, 4, 1, 8, false>(a, s, nm);
        case 11: return launch2_t<1, 4, 4, 1, 1, 4, 8, true>(a, s, nm);
        case 12: return launch2_t<1, 4, 8, 1, 1, 4, 8, true>(a, s, nm);
        case 13: return launch2_t<1, 9, 2, 1, 4, 1, 8, false>(a, s, nm);
        case 14: return launch2_t<1, 4, 2, 2, 4, 1, 8, false>(a, s, nm);
        case 15: return launch2_t<1, 4, 1, 2, 4, 1, 8, false>(a, s, nm);
        case 16: return launch2_t<1, 1, 2, 2, 4, 1, 8, false>(a, s, nm);
        case 17: return launch2_t<1, 1, 1, 2, 4, 1, 8, false>(a, s, nm);
        case 18: return launch2_t<1, 1, 2, 1, 4, 1, 8, false>(a, s, nm);
        case 19: return launch2_t<1, 4, 2, 1, 4, 1, 8, false>(a, s, nm);
        case 20: return launch2_t<2, 4, 2, 1, 4, 1, 8, false>(a, s, nm);
        case 21: return launch2_t<2, 4, 1, 2, 4, 1, 8, false>(a, s, nm);
        case 22: return launch2_t<1, 9, 4, 2, 4, 1, 8, false>(a, s, nm);
        case 23: return launch2_t<1, 1, 1, 2, 4, 1, 32, false>(a, s, nm);
        case 24: return launch2_t<1, 1, 2, 2, 2, 2, 32, false>(a, s, nm);
        case 25: return launch2_masked<1, 4, 2, 2, 4, 1, 8>(a, s, nm);
        case 26: return launch2_masked<1, 4, 1, 2, 4, 1, 8>(a, s, nm);
        case 27: return launch2_masked<1, 4, 2, 2, 4, 1, 16>(a, s, nm);
        case 28: return launch2_masked<1, 4, 1, 2, 4, 1, 16>(a, s, nm);
        case 29: return launch2_t<1, 9, 1, 1, 4, 1, 8, false>(a, s, nm);
    }
    set_error("conv2: bad variant %d", idx);
    return -3;
}

}  // namespace vfi
