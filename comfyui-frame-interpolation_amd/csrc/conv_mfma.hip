// NHWC fp32 convolution as an implicit GEMM on the gfx950 fp32 matrix cores.
//
//   M = output pixels (8x4-pixel sub-tiles of 32), N = output channels (sub-tiles of 32),
//   K = taps x input channels, contracted 2 at a time by v_mfma_f32_32x32x2_f32
//   (exact fp32: bit-for-bit an fmaf chain, so parity with the fp32 reference holds).
//
// Data movement per workgroup (256 threads = 4 wave64):
//   * activations: halo'd input tile, CK channels at a time, global -> registers -> LDS
//     (pixel stride CK+4 floats: ds_read_b128 conflict-free), next chunk prefetched into
//     registers while the current one is multiplied (issue-early / write-late);
//   * weights: host-packed [tap][Cin/8][Cout][8] so that one wave-wide global_load_dwordx4
//     (1 KiB contiguous, L2-resident) IS the B fragment of 4 consecutive MFMAs; they never
//     touch LDS and are prefetched one K-step ahead;
//   * epilogue fused: +bias, (*beta + residual), LeakyReLU, NHWC store (32 lanes = 128 B).
//
// Covers the reference's Conv2d(3x3, stride 1|2, pad 1) (+LeakyReLU), ResConv
// (vfi_models/rife/rife_arch.py:20-28,73-107) and — as 4 tap groups, one per output parity —
// ConvTranspose2d(4,2,1)+PixelShuffle(2) (rife_arch.py:215-218; index algebra SURVEY.md A7).
#include "vfi_common.h"

#include <atomic>

#include <cstdlib>
#include <map>
#include <string>
#include <cstring>

namespace vfi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// EXT: extended feature set (replicate padding, per-channel PReLU / sigmoid, post affine, interleaved
// transposed-conv store) compiled as a separate instantiation; the RIFE path runs EXT = false.
template <int STRIDE, int TAPS, int MT, int NT, int WM, int WN, int CK, bool GROUPED, bool EXT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(!GROUPED || (WN == 4 && NT == 1 && TAPS == 4), "grouped: one 2x2 tap group per wave column");
    static_assert(TAPS == 9 || TAPS == 4 || TAPS == 1, "tap rectangle 3x3 / 2x2 / 1x1");
    constexpr int KW = TAPS == 9 ? 3 : (TAPS == 4 ? 2 : 1);
    constexpr int SUBS = WM * MT;              // 8x4-pixel M sub-tiles per workgroup
    constexpr int SUBX = SUBS >= 2 ? 2 : 1;
    constexpr int SUBY = SUBS / SUBX;
    constexpr int TWO = SUBX * 8, THO = SUBY * 4;  // output tile
    constexpr int TWI = STRIDE * (TWO - 1) + 3;    // input tile incl. 1-px halo (taps in [-1,1])
    constexpr int THI = STRIDE * (THO - 1) + 3;
    constexpr int S = CK + 4;                      // LDS pixel stride in floats (S/4 odd)
    constexpr int NPIX = TWI * THI;
    constexpr int Q = CK / 4;                      // float4 per pixel per chunk
    constexpr int NITEM = NPIX * Q;
    constexpr int NLOAD = (NITEM + 255) / 256;
    constexpr int C8 = CK / 8;

    __shared__ __attribute__((aligned(16))) float lds[NPIX * S];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    // ---- which tile
    const int tiles_per_img = a.tiles_x * a.tiles_y;
    const int n = blockIdx.x / tiles_per_img;
    const int trem = blockIdx.x - n * tiles_per_img;
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int Y0 = ty * THO, X0 = tx * TWO;
    const int iy0 = STRIDE * Y0 - 1, ix0 = STRIDE * X0 - 1;

    const int g = GROUPED ? wn : 0;
    const int co0 = GROUPED ? blockIdx.y * 32 : (blockIdx.y * WN + wn) * NT * 32;
    const int cin8 = a.Cin_p >> 3;
    // tap rectangle origin: transposed-conv parity class g=(py,px) starts at (py-1, px-1)
    const int tby = GROUPED ? (g >> 1) - 1 : a.tap_y0;
    const int tbx = GROUPED ? (g & 1) - 1 : a.tap_x0;

    // ---- global -> register staging of the activation tile (geometry is chunk-invariant).
    // Loads are unconditional from a clamped address and zeroed by select: straight-line code.
    const int pstr = a.in_plane ? 4 : a.in_cs;                      // floats between pixels
    const int qstr = a.in_plane ? a.in_plane : 4;                   // floats between 4-channel groups
    const int cadv = a.in_plane ? (CK / 4) * a.in_plane : CK;       // floats between K chunks
    const float* in_n = a.in + (size_t)n * (a.in_plane ? (size_t)(a.in_cs / 4) * a.in_plane : (size_t)a.Hin * a.Win * a.in_cs);
    int goff[NLOAD];
    bool gok[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int idx = tid + i * 256;
        const int pix = idx / Q, q = idx - pix * Q;
        const int py = pix / TWI, px = pix - py * TWI;
        const int iy = iy0 + py, ix = ix0 + px;
        if (EXT && a.pad_replicate) {
            gok[i] = idx < NITEM;
            const int cy = min(max(iy, 0), a.Hin - 1), cx = min(max(ix, 0), a.Win - 1);
            goff[i] = gok[i] ? (cy * a.Win + cx) * pstr + q * qstr : 0;
        } else {
            gok[i] = idx < NITEM && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            goff[i] = gok[i] ? (iy * a.Win + ix) * pstr + q * qstr : 0;
        }
    }
    f32x4 stage[NLOAD];
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) stage[i] = *(const f32x4*)(in_n + goff[i] + (c0 / CK) * cadv);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = tid + i * 256;
            const int pix = idx / Q, q = idx - pix * Q;
            const f32x4 v = gok[i] ? stage[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            if ((i + 1) * 256 <= NITEM || idx < NITEM) *(f32x4*)&lds[pix * S + q * 4] = v;
        }
    };

    // ---- per-lane A fragment base offsets (floats) inside the LDS tile, tap origin folded in
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int s = wm * MT + mt;
        const int sx = s % SUBX, sy = s / SUBX;
        const int oy = sy * 4 + (l31 >> 3), ox = sx * 8 + (l31 & 7);
        abase[mt] = ((STRIDE * oy + 1 + tby) * TWI + STRIDE * ox + 1 + tbx) * S + half * 4;
    }

    // ---- per-lane B fragment pointer: packed [group][tap][Cin_p/8][Cout_p][8]
    const int tapstride = cin8 * a.Cout_p * 8;  // floats between taps
    const int c8stride = a.Cout_p * 8;          // floats between 8-channel K slices
    const float* wlane = a.w + (size_t)g * TAPS * tapstride + (co0 + l31) * 8 + half * 4;
    auto loadB = [&](f32x4(&b)[NT], const float* p) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = *(const f32x4*)(p + nt * 256);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    f32x4 bcur[NT], bnxt[NT];
    loadB(bcur, wlane);
    gload(0);
    for (int c0 = 0; c0 < a.Cin_p; c0 += CK) {
        __syncthreads();  // every wave is done reading the previous chunk
        lstore();
        __syncthreads();
        const bool has_next = c0 + CK < a.Cin_p;
        if (has_next) gload(c0 + CK);
        const float* wchunk = wlane + (c0 >> 3) * c8stride;
        // B fragment after this chunk's last K-step: first step of the next chunk (or a harmless reload)
        const float* wwrap = has_next ? wchunk + C8 * c8stride : wlane;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int toff = ((t / KW) * TWI + (t % KW)) * S;
#pragma unroll
            for (int c8 = 0; c8 < C8; ++c8) {
                const bool last = (t == TAPS - 1) && (c8 == C8 - 1);
                const float* pn = last ? wwrap
                                       : wchunk + (c8 + 1 == C8 ? (t + 1) * tapstride : t * tapstride + (c8 + 1) * c8stride);
                loadB(bnxt, pn);
                f32x4 av[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = *(const f32x4*)&lds[abase[mt] + toff + c8 * 8];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mt][j], bcur[nt][j],
                                                                               acc[mt][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bcur[nt] = bnxt[nt];
            }
        }
    }

    // ---- epilogue: lane holds output channel co for 16 pixels of each sub-tile
    // Fast paths chosen once per workgroup (uniform): the generic loop below re-tests act / res / beta / out_mode for every
    // one of the 16*MT*NT values of a lane (hundreds of scalar branches per tile).
    const bool plain = a.res == nullptr && (a.act == 0 || a.act == 1 || (EXT && a.act == 3)) && !(EXT && a.post_scale != 0.f);
    const bool nhwc = a.out_mode == 0 && !GROUPED;
    const bool inter = EXT && GROUPED && a.out_mode == 2;       // transposed conv, parity groups interleaved into NHWC
    if (plain && (nhwc || inter)) {
        const int m = inter ? 2 : 1;
        const int xstr = m * a.out_cs, ystr = m * m * a.Wout * a.out_cs;
        const bool interior = Y0 + (SUBS / SUBX) * 4 <= a.Hout && X0 + SUBX * 8 <= a.Wout;
        const bool unif01 = a.act == 0 || (a.act == 1 && a.slope >= 0.f && a.slope <= 1.f);
        const float uslope = a.act == 1 ? a.slope : 1.0f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = co0 + nt * 32 + l31;
            if (co >= a.Cout) continue;
            const float bs = a.bias[g * a.Cout_p + co], bt = a.beta ? a.beta[co] : 1.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int s = wm * MT + mt;
                const int sx = s % SUBX, sy = s / SUBX;
                const int oy0 = Y0 + sy * 4, ox0 = X0 + sx * 8 + 4 * half;
                float* ob = a.out + ((size_t)(n * m * a.Hout + m * oy0 + (inter ? (g >> 1) : 0)) * (m * a.Wout) + m * ox0 +
                                     (inter ? (g & 1) : 0)) * a.out_cs + co;
                if (unif01 && interior) {   // the common case: uniform slope in [0,1] -> lrelu(v) == max(v, v*slope)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = (acc[mt][nt][r] + bs) * bt;
                        ob[(r >> 2) * ystr + (r & 3) * xstr] = fmaxf(v, v * uslope);
                    }
                } else {
                    const float sl = a.act == 0 ? 1.0f : (a.act == 1 ? a.slope : a.prelu[co]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = (acc[mt][nt][r] + bs) * bt;
                        v = v > 0.f ? v : v * sl;
                        if (interior || (oy0 + (r >> 2) < a.Hout && ox0 + (r & 3) < a.Wout)) ob[(r >> 2) * ystr + (r & 3) * xstr] = v;
                    }
                }
            }
        }
        return;
    }
    if (!EXT && plain && GROUPED && a.out_mode == 1 && a.act == 0 && a.beta == nullptr) {
        // transposed conv + PixelShuffle(2): value (parity group g, channel co) of input pixel (oy, ox) goes to plane c/4,
        // component c%4 (c = co/4) of pixel (4*oy + 2*gy + (co>>1)&1, 4*ox + 2*gx + co&1) of the planar4 output
        const int Ws = 4 * a.Wout, Hs = 4 * a.Hout;
        const int planes = a.out_planes ? a.out_planes : 2;
        const bool interior = Y0 + (SUBS / SUBX) * 4 <= a.Hout && X0 + SUBX * 8 <= a.Wout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = co0 + nt * 32 + l31;
            if (co >= a.Cout) continue;
            const float bs = a.bias[g * a.Cout_p + co];
            const int c = co >> 2;
            float* lane_base = a.out + ((size_t)(n * planes + (c >> 2)) * Hs * Ws + (size_t)(2 * (g >> 1) + ((co >> 1) & 1)) * Ws +
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
                        ob[((r >> 2) * 4 * Ws + (r & 3) * 4) * 4] = acc[mt][nt][r] + bs;
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = co0 + nt * 32 + l31;
        const bool cok = co < a.Cout;
        const float bs = a.bias[g * a.Cout_p + co];
        const float bt = a.beta ? a.beta[co] : 1.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int s = wm * MT + mt;
            const int sx = s % SUBX, sy = s / SUBX;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oy = Y0 + sy * 4 + (r >> 2);
                const int ox = X0 + sx * 8 + (r & 3) + 4 * half;
                if (cok && oy < a.Hout && ox < a.Wout) {
                    const size_t p = (size_t)(n * a.Hout + oy) * a.Wout + ox;
                    float v = acc[mt][nt][r] + bs;
                    if (a.beta) v *= bt;
                    if (a.res) v += a.res[p * a.res_cs + co];
                    if (a.act == 1) v = v > 0.f ? v : v * a.slope;
                        else if (a.act == 2) v = fminf(fmaxf(v, 0.f), 1.f);
                        else if (EXT && a.act == 3) v = v > 0.f ? v : v * a.prelu[co];
                        else if (EXT && a.act == 4) v = 1.0f / (1.0f + expf(-v));
                        else if (EXT && a.act == 5) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                        if (EXT && a.post_scale != 0.f) v = v * a.post_scale + a.post_shift;
                    if (GROUPED && a.out_mode == 1) {
                        const int Ws = 4 * a.Wout, Hs = 4 * a.Hout, c = co >> 2;
                        const int Yt = 4 * oy + 2 * (g >> 1) + ((co >> 1) & 1), Xt = 4 * ox + 2 * (g & 1) + (co & 1);
                        a.out[((size_t)(n * (a.out_planes ? a.out_planes : 2) + (c >> 2)) * Hs * Ws + (size_t)Yt * Ws + Xt) * 4 + (c & 3)] = v;
                    } else if (EXT && GROUPED && a.out_mode == 2) {
                        const size_t q2 = (size_t)(n * 2 * a.Hout + 2 * oy + (g >> 1)) * (2 * a.Wout) + 2 * ox + (g & 1);
                        a.out[q2 * a.out_cs + co] = v;
                    } else {
                        a.out[p * a.out_cs + g * a.Cout_p + co] = v;
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// variant table
// ------------------------------------------------------------------------------------------
static const ConvVariant kVariants[] = {
    // name          stride taps mt nt wm wn ck grouped
    {"s1_m2n2", 1, 9, 2, 2, 4, 1, 16, 0},     // 0: 16x16 px x 64 ch
    {"s1_m2n3", 1, 9, 2, 3, 4, 1, 16, 0},     // 1: 16x16 px x 96 ch
    {"s1_m1n2", 1, 9, 1, 2, 4, 1, 16, 0},     // 2: 16x8 px x 64 ch
    {"s1_m1n3", 1, 9, 1, 3, 4, 1, 16, 0},     // 3: 16x8 px x 96 ch
    {"s1_m1n1", 1, 9, 1, 1, 4, 1, 16, 0},     // 4: 16x8 px x 32 ch
    {"s1_m2n1", 1, 9, 2, 1, 4, 1, 16, 0},     // 5: 16x16 px x 32 ch
    {"s1_m1n1w22", 1, 9, 1, 1, 2, 2, 16, 0},  // 6: 16x4 px x 64 ch
    {"s1_m2n2w22", 1, 9, 2, 2, 2, 2, 16, 0},  // 7: 16x8 px x 128 ch
    {"s2_m1n2", 2, 9, 1, 2, 4, 1, 8, 0},      // 8: stride 2, 16x8 px x 64 ch
    {"s2_m1n3", 2, 9, 1, 3, 4, 1, 8, 0},      // 9: stride 2, 16x8 px x 96 ch
    {"s2_m1n1", 2, 9, 1, 1, 4, 1, 8, 0},      // 10: stride 2, 16x8 px x 32 ch
    {"s2_m2n2", 2, 9, 2, 2, 4, 1, 8, 0},      // 11: stride 2, 16x16 px x 64 ch
    {"g_m4", 1, 4, 4, 1, 1, 4, 16, 1},        // 12: 4 parity groups of 2x2 taps, 16x8 px
    {"g_m2", 1, 4, 2, 1, 1, 4, 16, 1},        // 13: same, 16x4 px
};
int conv_num_variants() { return (int)(sizeof(kVariants) / sizeof(kVariants[0])); }
const ConvVariant& conv_variant(int i) { return kVariants[i]; }
const ConvVariant* conv_variant_lookup(int id) {
    if (id >= 0 && id < conv_num_variants()) return &kVariants[id];
    if (id >= kConv2Base && id < kConv2Base + conv2_num_variants()) return &conv2_variant(id - kConv2Base);
    return nullptr;
}

template <int STRIDE, int TAPS, int MT, int NT, int WM, int WN, int CK, bool GROUPED>
static int launch_t(ConvArgs a, hipStream_t s, const char* name) {
    constexpr int SUBS = WM * MT;
    constexpr int SUBX = SUBS >= 2 ? 2 : 1;
    constexpr int SUBY = SUBS / SUBX;
    a.tiles_x = cdiv(a.Wout, SUBX * 8);
    a.tiles_y = cdiv(a.Hout, SUBY * 4);
    VFI_REQUIRE(a.Cin_p % CK == 0, "conv %s: Cin_p=%d not a multiple of the K chunk %d", name, a.Cin_p, CK);
    VFI_REQUIRE(a.Cout_p % (GROUPED ? 32 : WN * NT * 32) == 0, "conv %s: Cout_p=%d not a multiple of the N tile %d",
                name, a.Cout_p, WN * NT * 32);
    dim3 grid(a.N * a.tiles_x * a.tiles_y, GROUPED ? a.Cout_p / 32 : a.Cout_p / (WN * NT * 32));
    TraceScope ts(name, s);
    if (a.pad_replicate || a.act >= 3 || a.post_scale != 0.f || a.out_mode == 2)
        hipLaunchKernelGGL((conv_mfma_kernel<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<STRIDE, TAPS, MT, NT, WM, WN, CK, GROUPED, false>), grid, dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

static int n_cus_cached() {
    static std::atomic<int> n_of[kMaxDevices];     // per device: one process may drive several
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int n = n_of[dev].load(std::memory_order_relaxed);
    if (!n) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
        n_of[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Tile-shape heuristic, from tools/conv_sweep.py on MI355X (profiles/r01_conv_sweep_*.txt):
//   * stride 1: second-generation (LDS-DMA, persistent) 16x16x64 tile for 64 channels, 16x8x64 for the
//     other multiples of 64, 16x8x96 for 96 channels;
//   * stride 2 with Cout a multiple of 64: second generation 16x8x64; the other stride-2 shapes and the
//     grouped transposed conv: first-generation small tiles.
int conv_pick_variant(const ConvArgs& a, int stride, bool grouped) {
    // The layer objects (split_ok: FILM / M2M / ..., batch fixed by the model) size the choice by the launch; the RIFE network, whose
    // launches carry 1..32 tasks, by the image at a nominal 8 tasks: tile variants differ in K-chunk size, i.e. in fp32 summation
    // order, and a frame's result must not depend on how many tasks shared its launch (tests/test_gpu_rife.py: bit-equal).
    const long px = (long)(a.split_ok ? a.N : 8) * a.Hout * a.Wout;
    // up-sample x2 + 2x2 (tap masks): the 16x8x64 tile with 16-channel chunks — 2 .. 8 K-steps per barrier where the 8-channel chunk has 1 .. 4
    // (tools/film_variant_ab.py at 1080p, ms for FILM's three layers: m2n2 2.40, m1n2 2.26, m2n2k16 2.27, m1n2k16 2.07)
    if (!grouped && stride == 1 && a.ntaps == 4 && a.tapmask) return kConv2Base + (a.Cin_p % 16 == 0 ? 28 : 26);
    if (!grouped && stride == 2 && a.ntaps == 4) return kConv2Base + (a.Cout_p % 64 == 0 ? 21 : 20);
    if (!grouped && stride == 1 && a.ntaps != 9) {  // 2x2 'same' / 1x1 convs (FILM): second generation only
        const bool big = px * (a.Cout_p / 32) >= 256L * 4 * n_cus_cached();
        // 1x1: a K chunk of 8 channels is ONE K-step per barrier; with 32-channel chunks (4 steps) GMFlow's projections / FFN
        // run 10-25 % faster (tools/conv1x1_bench.py, profiles/r02b_conv1x1_variants.txt); wide inputs take the 128-channel N tile
        if (a.ntaps == 1 && a.Cout_p % 64 == 0 && a.Cin_p % 32 == 0)
            return kConv2Base + (a.Cin_p >= 512 && a.Cout_p % 128 == 0 ? 24 : 23);     // d1t1_m2n2w22k32 / d1t1_m1n2k32
        if (a.Cout_p % 64 == 0) return kConv2Base + (a.ntaps == 4 ? (big ? 14 : 15) : (big ? 16 : 17));
        return kConv2Base + (a.ntaps == 4 ? 19 : 18);
    }
    const long cus = n_cus_cached();
    if (grouped) {  // big pixel-shuffled block outputs (RIFE lastconv of blocks 2/3): LDS-DMA variant dg_m4 (8 % faster there)
        if (a.out_mode == 1 && px >= 250000) return kConv2Base + 11;
        return a.Cin_p % 16 == 0 ? 13 : kConv2Base + 11;
    }
    const bool n3 = a.Cout_p % 96 == 0;
    const bool n2 = a.Cout_p % 64 == 0;
    (void)cus;
    if (stride == 2) {
        if (n2) return kConv2Base + 7;   // d2_m1n2
        if (n3) return kConv2Base + 9;   // d2_m1n3 (r6: 0.924 -> 0.832 ms on RIFE's conv0b_b2 against the first-generation s2_m1n3, same bits)
        return 10;                       // s2_m1n1
    }
    // coarse pyramid levels of the layer objects (M2M's PWC decoders at 17x30 / 34x60: 24 .. 80 workgroups of the 64-channel tile on 256
    // CUs, each walking the whole K loop alone — 46 us for 120 -> 128 @17x30): the 32-channel N tile doubles the workgroups and halves
    // each one's serial MFMA chain; same K chunk (8), i.e. the same summation order and bits as d1_m1n2
    if (a.split_ok && (n2 || n3) && px <= 6000) return kConv2Base + 29;   // d1_m1n1
    if (n2) {   // d1_m2n2 (16x16 px tile) / d1_m1n2 (16x8): the larger M tile re-uses each weight fragment twice as often
        // A/B option m2n2_px: pixel count from which wide layers take m2n2.  Measured on FILM / M2M at 1080p
        // (profiles/r02_film_tile_experiment.txt): 100k is the best of {never, 1.5M, 400k, 100k, 20k} by 0.6 % only — not
        // worth sharing the block-3 ResConv's kernel instantiation with other layers, so the default stays "never"
        const long big_px = option(kOptM2n2Px);
        if (a.Cout_p == 64 || (big_px >= 0 && px >= big_px)) return kConv2Base + 0;
        return kConv2Base + 2;
    }
    if (n3) return px >= 100000 || a.Cin_p % 16 ? kConv2Base + 3 : 4;  // d1_m1n3 / s1_m1n1 (K chunk 16)
    return a.Cin_p % 16 == 0 && px < 20000 ? 4 : kConv2Base + 13;      // 32-channel N tile: s1_m1n1 / d1_m2n1
}

int conv_launch(const ConvArgs& a, int stride, bool grouped, int variant, hipStream_t s,
                const char* trace_name) {
    if (variant < 0 && grouped && option(kOptGroupedVariant) >= 0) variant = (int)option(kOptGroupedVariant);   // A/B option grouped_variant
    if (variant < 0 && trace_name) variant = variant_override(trace_name);      // A/B hook vfi_test_variant_override (by trace name)
    if (variant < 0) variant = conv_pick_variant(a, stride, grouped);
    const ConvVariant* vp = conv_variant_lookup(variant);
    VFI_REQUIRE(vp, "conv: bad variant %d", variant);
    const ConvVariant& v = *vp;
    VFI_REQUIRE(v.stride == stride && (v.grouped != 0) == grouped && v.taps == a.ntaps,
                "conv: variant %s does not match stride %d grouped %d taps %d", v.name, stride, (int)grouped,
                a.ntaps);
    VFI_REQUIRE(a.out_mode != 1 || (grouped && a.Cout % 4 == 0 && a.Cout / 4 <= 4 * (a.out_planes ? a.out_planes : 2)),
                "conv: out_mode 1 is for grouped convs whose pixel-shuffled channels fit the planar4 output");
    VFI_REQUIRE(a.out_mode != 2 || grouped, "conv: out_mode 2 is for grouped convs");
    VFI_REQUIRE(a.act != 3 || a.prelu, "conv: act 3 needs per-channel slopes");
    VFI_REQUIRE(a.Cin_p % 8 == 0 && a.Cout_p % 32 == 0 && a.in_cs >= a.Cin_p && a.in_cs % 4 == 0,
                "conv: bad channel padding Cin_p=%d Cout_p=%d in_cs=%d", a.Cin_p, a.Cout_p, a.in_cs);
    VFI_REQUIRE(((uintptr_t)a.in & 15) == 0 && ((uintptr_t)a.w & 15) == 0, "conv: unaligned pointers");
    const char* nm = trace_name ? trace_name : v.name;
    if (variant >= kConv2Base) return conv2_launch(a, variant - kConv2Base, s, nm);
    switch (variant) {
        case 0: return launch_t<1, 9, 2, 2, 4, 1, 16, false>(a, s, nm);
        case 1: return launch_t<1, 9, 2, 3, 4, 1, 16, false>(a, s, nm);
        case 2: return launch_t<1, 9, 1, 2, 4, 1, 16, false>(a, s, nm);
        case 3: return launch_t<1, 9, 1, 3, 4, 1, 16, false>(a, s, nm);
        case 4: return launch_t<1, 9, 1, 1, 4, 1, 16, false>(a, s, nm);
        case 5: return launch_t<1, 9, 2, 1, 4, 1, 16, false>(a, s, nm);
        case 6: return launch_t<1, 9, 1, 1, 2, 2, 16, false>(a, s, nm);
        case 7: return launch_t<1, 9, 2, 2, 2, 2, 16, false>(a, s, nm);
        case 8: return launch_t<2, 9, 1, 2, 4, 1, 8, false>(a, s, nm);
        case 9: return launch_t<2, 9, 1, 3, 4, 1, 8, false>(a, s, nm);
        case 10: return launch_t<2, 9, 1, 1, 4, 1, 8, false>(a, s, nm);
        case 11: return launch_t<2, 9, 2, 2, 4, 1, 8, false>(a, s, nm);
        case 12: return launch_t<1, 4, 4, 1, 1, 4, 16, true>(a, s, nm);
        case 13: return launch_t<1, 4, 2, 1, 1, 4, 16, true>(a, s, nm);
    }
    return -3;
}

// ------------------------------------------------------------------------------------------
// host-side packing
// ------------------------------------------------------------------------------------------
void conv3x3_taps(ConvArgs& a) {
    a.ntaps = 9;
    a.tap_y0 = a.tap_x0 = -1;
}
void deconv4x4_taps(ConvArgs& a) {
    a.ntaps = 4;  // per parity group (py,px): 2x2 taps with origin (py-1, px-1), derived in the kernel
    a.tap_y0 = a.tap_x0 = 0;
}

void pack_conv3x3(const float* w, const float* bias, int Cout, int Cin, int Cin_p, int Cout_p,
                  std::vector<float>& wp, std::vector<float>& bp) {
    const int cin8 = Cin_p / 8;
    wp.assign((size_t)9 * cin8 * Cout_p * 8, 0.f);
    bp.assign(Cout_p, 0.f);
    for (int co = 0; co < Cout; ++co) {
        if (bias) bp[co] = bias[co];
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < 9; ++t)
                wp[(((size_t)t * cin8 + ci / 8) * Cout_p + co) * 8 + (ci & 7)] =
                    w[((size_t)co * Cin + ci) * 9 + t];
    }
}

void pack_deconv4x4(const float* w, const float* bias, int Cin, int Cout, int Cin_p, int Cout_p,
                    std::vector<float>& wp, std::vector<float>& bp) {
    // out[co, 2y+py, 2x+px] = b[co] + sum_{ci, a, b} in[ci, y+dy, x+dx] * w[ci, co, ky, kx]
    //   dy = py-1+a, ky = 3 - py - 2a   (from oy = 2*iy - 1 + ky)
    const int cin8 = Cin_p / 8;
    wp.assign((size_t)4 * 4 * cin8 * Cout_p * 8, 0.f);
    bp.assign((size_t)4 * Cout_p, 0.f);
    for (int g = 0; g < 4; ++g) {
        const int py = g >> 1, px = g & 1;
        for (int co = 0; co < Cout; ++co) {
            if (bias) bp[(size_t)g * Cout_p + co] = bias[co];
            for (int t = 0; t < 4; ++t) {
                const int ky = 3 - py - 2 * (t >> 1), kx = 3 - px - 2 * (t & 1);
                for (int ci = 0; ci < Cin; ++ci)
                    wp[((((size_t)g * 4 + t) * cin8 + ci / 8) * Cout_p + co) * 8 + (ci & 7)] =
                        w[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// cross-check kernel: one thread per output element, plain fmaf loop
// ------------------------------------------------------------------------------------------
__global__ void conv3x3_naive_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                     const float* __restrict__ bias, const float* __restrict__ beta,
                                     float* __restrict__ out, int N, int H, int W, int Cin, int in_cs,
                                     int Cout, int stride, int Ho, int Wo, int act, float slope) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)N * Ho * Wo * Cout;
    if (idx >= total) return;
    const int co = idx % Cout;
    long p = idx / Cout;
    const int ox = p % Wo;
    p /= Wo;
    const int oy = p % Ho;
    const int n = p / Ho;
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * stride - 1 + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * stride - 1 + kx;
            if (ix < 0 || ix >= W) continue;
            const float* ip = in + ((size_t)(n * H + iy) * W + ix) * in_cs;
            const float* wp = w + (size_t)co * Cin * 9 + ky * 3 + kx;
            for (int ci = 0; ci < Cin; ++ci) acc = fmaf(ip[ci], wp[(size_t)ci * 9], acc);
        }
    }
    float v = acc + (bias ? bias[co] : 0.f);
    if (beta) v = v * beta[co] + in[((size_t)(n * H + oy) * W + ox) * in_cs + co];
    if (act == 1) v = v > 0.f ? v : v * slope;
    out[idx] = v;
}

int conv_naive_launch(const float* in, const float* w_dev, const float* bias_dev,
                      const float* beta_dev, float* out, int N, int H, int W, int Cin, int in_cs,
                      int Cout, int stride, int act, float slope, hipStream_t s) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long total = (long)N * Ho * Wo * Cout;
    TraceScope ts("conv3x3_naive", s);
    hipLaunchKernelGGL(conv3x3_naive_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in,
                       w_dev, bias_dev, beta_dev, out, N, H, W, Cin, in_cs, Cout, stride, Ho, Wo, act,
                       slope);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vfi
