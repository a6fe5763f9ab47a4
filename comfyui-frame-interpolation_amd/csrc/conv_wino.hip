// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores of gfx950 (stride 1, pad 1; NHWC fp32).
//
// Replaces the 3x3 stride-1 convolutions of the reference — ResConv in IFBlock (vfi_models/rife/rife_arch.py:20-28,237-276),
// FILM's 'same' 3x3 convs (film_arch.py:784-798), the M2M / IFRNet decoders — which the direct implicit-GEMM kernel
// (conv_mfma2.hip) already runs at 0.84 of the fp32-MFMA peak: the remaining lever is the multiplication count.
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      (Lavin & Gray; 4x4 input patch d -> 2x2 outputs, 16 products instead of 36)
// turns the convolution into 16 independent GEMMs  M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[ci][co]  (xi = 0..15), i.e. 2.25x
// fewer v_mfma_f32_32x32x2_f32 per output than the direct form, at identical (fp32, fmaf-chain) accumulation.
//
// MI355X mapping (one wave per SIMD, 512 registers per lane):
//   * a wave owns a REGION of 32 tiles (8x4 tiles = 16x8 output pixels, or 16x2 = 32x4 for small images) and 32 output
//     channels: 16 accumulators of 32x32 (tile x channel) = 256 AGPRs stay resident through the whole K loop;
//   * the input transform B^T d B runs in registers, in EXACTLY the MFMA A-operand layout: lane (m = tile, half) reads its
//     4x4 patch for channels 4*half .. 4*half+3 with 16 ds_read_b128 (LDS image [pixel][8 ch], LDS-DMA'd from the NHWC
//     activation, out-of-image pixels zero-filled by the buffer descriptor), 32 v_add/v_sub per channel give the 16 V_xi
//     values = the A operands of 16 MFMAs.  No transformed-input tensor ever exists in LDS or HBM;
//   * U = G g G^T is computed on the host at weight-pack time, laid out [Cout/32][Cin/8][j][xi/4][half][co][xi%4] so that a
//     (j, xi/4) slice is one 1 KiB LDS-DMA piece whose lane-linear image IS the B-operand order: one ds_read_b128 feeds the
//     B operands of 4 MFMAs;
//   * the output transform A^T M A runs in registers on the accumulator layout (lane = output channel, register = tile),
//     fused with bias, beta, residual, activation and the NHWC store (32 lanes = 128 contiguous bytes per pixel);
//   * K pipeline: 8-channel chunks, three LDS buffers, the DMA queue three chunks ahead across work-item boundaries.  A wave's own
//     VALU work is NOT hidden by its MFMAs (one wave per SIMD: ~15 cycles of matrix-pipe time per VALU burst + ~4 per
//     instruction, tools/micro/mfma_shadow.hip — LDS reads, DMA issue and SALU are free), so the transform runs in packed fp32
//     (v_pk_add_f32 over channel pairs: half the instructions) in four bursts per chunk, operands ping-pong by name (no
//     copies), and the one barrier per chunk sits only in front of the shared weight reads;
//   * regions are wave-private (each wave DMA's its own halo'd patch: only the weight tile is shared by the workgroup), so
//     image sizes quantise to 16x8 (or 32x4) pixels instead of a 4-wave tile;
//   * persistent workgroups, XCD-aware work order: the Cout/32 siblings of one region quad run on the SAME XCD at the same
//     time (block b -> XCD b % 8), so the activation is fetched from HBM once and hit in that XCD's L2 by the siblings.
#include "vfi_common.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/vfi_hip.h"
#include "../../include/vfi_hip_test.h"

namespace vfi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

struct WinoArgs {
    ConvArgs a;     // in / bias / beta / res / prelu / out, sizes, act ... (a.w = the Winograd-packed weights)
    int rx, ry;     // regions per image
    int R;          // regions in total (N * ry * rx)
    int NQ;         // region quads (4 regions = the 4 waves of a workgroup)
    int NY;         // Cout_p / 32
    int xcd_map;    // 1: XCD-aware work order (gridDim.x % 8 == 0)
    float inv_NY, inv_per, inv_rx;   // 1 / NY, 1 / (rx * ry), 1 / rx: the work decode divides by multiplication (see wino_div)
    int LO;                          // SHUF epilogue: channels per parity group (Cout / 4) and
    float inv_LO;                    //   its reciprocal (an integer division in the epilogue keeps a hoisted reciprocal alive — spilled — through the K loop)
    int clk_tag;                     // clock probe: the host's index of this launch (read at the kernel's first instructions only)
};

// Clock probe state of the device (vfi_clock_probe): record buffer, capacity, next free record.  Device globals, not kernel arguments:
// a pointer argument that is used again at the kernel's end stays live through the K loop and cost nine more spilled SGPRs there.
__device__ unsigned long long* g_clk_buf_dev = nullptr;
__device__ unsigned g_clk_cap_dev = 0, g_clk_next_dev = 0;
constexpr int kClkRecWords = 8;      // u64 per record: t0, r0, t1, r1, tag, 0, 0, 0 (one 64-byte line)

// Cycle ledger of the K loop (test option wino_probe, docs/design/winograd.md "cycle ledger"): the PROBE instantiations of the hot kernel
// sum s_memtime stamps (mod 2^32) taken at fixed points of every chunk / item by the waves of workgroup 0; differences of the sums are
// the cycles spent between the points.  Each stamp is consumed at the NEXT point where the loop waits lgkmcnt(0) anyway (s_memtime
// returns through lgkmcnt, out of order with LDS reads: consuming it earlier would add a full LDS drain to the thing being measured),
// and one PROBE value takes at most four stamps per chunk (SGPR pressure).  [wave][0..7] sums, written at the kernel's end.
__device__ unsigned g_wino_probe_out[4][8];

template <int RTX>
struct WinoGeom {
    static constexpr int RTY = 32 / RTX;
    static constexpr int RW = 2 * RTX, RH = 2 * RTY;     // output pixels of a region
    static constexpr int PW = RW + 2, PH = RH + 2;       // its input patch
    static constexpr int NPIX = PW * PH;
    static constexpr int NITEM = NPIX * 2;               // 16-byte items per region per 8-channel chunk
    static constexpr int NA = (NITEM + 63) / 64;         // 1 KiB DMA pieces per wave and chunk
    static constexpr int A_FLOATS = NA * 256;
    static constexpr int B_FLOATS = 16 * 256;
    static constexpr int BUF_FLOATS = 4 * A_FLOATS + B_FLOATS;
    static constexpr int TAB_FLOATS = 4 * NA * 64;       // per wave: the DMA cursor's NA byte offsets per lane (kept in LDS, not in VGPRs)
    static constexpr int NBUF = 3;                       // LDS chunk buffers: the DMA queue runs two chunks ahead of the MFMAs
    static constexpr int MAXCO = 1024;                   // output channels whose epilogue constants (bias, beta, PReLU slope) sit in LDS
    static constexpr int LDS_BYTES = (NBUF * BUF_FLOATS + TAB_FLOATS + 3 * MAXCO + 4) * 4;      // + the clock probe's record index
    static_assert(NITEM % 4 == 0 && PW % 2 == 0, "swizzle stays inside the image and inside a row");
};

// LDS slot (16-byte units) of item (pixel (py, px), channel quad q) inside a wave's A image: the natural slot with its low two
// bits XORed by a row-pair key.  A wave's ds_read_b128 of one patch position touches tiles 2 px apart = slots 4 apart, i.e.
// only every fourth 16-byte bank group; the key spreads the four tile rows over the four groups (conflict-free for 8x4 tiles).
__device__ __forceinline__ int wino_key(int py) { return (py >> 1) & 3; }

// s_waitcnt immediate of gfx9 / CDNA: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14 (expcnt 7, lgkmcnt 15 = no wait).
// The builtin, not inline asm: hipcc's own wait-count pass then KNOWS the counters' state (behind an opaque asm it assumed the
// B-fragment / offset reads of the previous sub-step could still be outstanding and stalled the boundary's first MFMA / first
// DMA piece on the oldest of the 16 patch reads issued in between).
constexpr int wino_waitcnt(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

// n / d for 0 <= n < 2^24 with the host's 1.0f / d: float product, one fix-up step.  (hipcc's own integer division keeps a hoisted
// reciprocal in a VGPR for the whole kernel — one more register to spill, and its reload in the per-item code sat behind
// s_waitcnt vmcnt(0), i.e. behind all LDS-DMA in flight.)
__device__ __forceinline__ int wino_div(int n, int d, float inv_d) {
    int q = (int)((float)n * inv_d);
    const int r = n - q * d;
    q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
    return __builtin_amdgcn_readfirstlane(q);      // wave-uniform by construction; tell hipcc (the float ops run on the VALU)
}


// v_pk_*_f32 by hand: LLVM scalarises every <2 x float> fsub (and folds fma(b, -1, a) back into one)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 wino_pk_add(f32x2 x, f32x2 y) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ f32x2 wino_pk_sub(f32x2 x, f32x2 y) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
// x + (c.lo, c.lo) and x * (c.hi, c.hi): one register pair (bias, beta) serves both lanes of both operations (op_sel picks the half)
__device__ __forceinline__ f32x2 wino_pk_add_lo(f32x2 x, f32x2 c) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 wino_pk_mul_hi(f32x2 x, f32x2 c) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 wino_pk_mul(f32x2 x, f32x2 y) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

// Output transform + fused epilogue of one region (one wave).  acc[xi][r]: lane = output channel (l31), register r = tile
// m = 8 * (r >> 2) + 4 * half + (r & 3) (MFMA 32x32 D layout).  Stores and residual loads go through buffer descriptors of the
// image: an out-of-image pixel (or a padded output channel) gets offset 0x80000000 and the hardware drops / zero-fills it —
// no per-value branches.  MODE 0: the hot form (no residual, none / LeakyReLU with a slope in [0,1]: lrelu(v) = max(v, v*slope));
// MODE 1 (r5): the hot form with a PER-CHANNEL negative slope (PReLU(c): v > 0 ? v : v * slope[co] — the lane IS the output channel, so
// the slope is one register; M2M's / IFUNet's conv + PReLU and ConvTranspose2d + PReLU layers, which used to take the general form);
// MODE 10 + act: the general form (residual, any activation of ConvArgs::act, post affine).
// FULL: the region lies completely inside the image (wave-uniform, true for all but the last row / column of regions): no
// row branches and no per-store column test (they were ~260 of the epilogue's 1350 instructions).
// SHUF != 0: the layer is a ConvTranspose2d(4, 2, 1) written as a 3x3 convolution — output channel co = parity group g x LO + cg
// (pack_deconv_as_conv3x3) — and the epilogue puts the parities where they belong:
//   1 (ConvArgs::out_mode 1, RIFE lastconv): + PixelShuffle(2) into the planar4 block output T [n][planes][4H][4W][4]: pixel
//     (4 oy + 2 (g >> 1) + ((cg >> 1) & 1), 4 ox + 2 (g & 1) + (cg & 1)), plane (cg >> 2) >> 2, component (cg >> 2) & 3 — the mapping
//     of conv_mfma2's out_mode 1 epilogue;
//   2 (out_mode 2, the layer objects of M2M / IFRNet / IFUNet / GMFSS): NHWC [n][2H][2W][out_cs], pixel (2 oy + (g >> 1), 2 ox + (g & 1)),
//     channel cg.  No residual in either form.
template <int RTX, int MODE, bool FULL = false, int SHUF = 0>
__device__ __forceinline__ void wino_epilogue(const f32x16 (&acc)[16], const ConvArgs& a, int n, int oy0, int ox0, int co, int coc, int half, float bs,
                                              float bt, float pre, int LO = 0, float inv_LO = 0.f) {
    const int H = a.Hin, W = a.Win;
    const float uslope = a.act == 1 ? a.slope : 1.0f;
    const float ps = a.post_scale != 0.f ? a.post_scale : 1.0f, sh = a.post_scale != 0.f ? a.post_shift : 0.0f;
    const int planes = a.out_planes ? a.out_planes : 2;
    const int Ws4 = 4 * W;                       // SHUF: T row length in pixels
    const __amdgpu_buffer_rsrc_t orsrc =
        SHUF == 1 ? __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * planes * 16 * H * W * 4), 0, planes * 16 * H * W * 16, 0x00020000)
        : SHUF == 2 ? __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * 4 * H * W * a.out_cs), 0, 4 * H * W * a.out_cs * 4, 0x00020000)
                    : __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * H * W * a.out_cs), 0, H * W * a.out_cs * 4, 0x00020000);
    const bool has_res = MODE >= 10 && a.res != nullptr;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_res ? a.res + (size_t)n * H * W * a.res_cs : a.out), 0, has_res ? H * W * a.res_cs * 4 : 0, 0x00020000);
    const int xlane = ox0 + 8 * half;          // this lane's first output column; + xq per store
    const bool cok = co < a.Cout;
    const bool interior = FULL || ox0 + 2 * RTX <= W;  // every column of the region is inside the image (wave-uniform)
    // lane part of the byte offsets in a VGPR (0x80000000 = dropped by the descriptor's range check), per-store part scalar
    int lane_o = cok ? (xlane * a.out_cs + co) * 4 : (int)0x80000000;
    const int lane_r = cok ? (xlane * a.res_cs + co) * 4 : (int)0x80000000;
    if (SHUF && cok) {      // lane part: plane, sub-pixel of the 4x4 cell, component, and the lane's first column
        int g = (int)((float)co * inv_LO), cg = co - g * LO;      // co / LO by multiplication + one fix-up (0 <= co < 2^10)
        if (cg < 0) --g, cg += LO;
        if (cg >= LO) ++g, cg -= LO;
        const int c = cg >> 2;
        if (SHUF == 1) lane_o = ((((c >> 2) * 4 * H + 2 * (g >> 1) + ((cg >> 1) & 1)) * Ws4 + 2 * (g & 1) + (cg & 1) + 4 * xlane) * 4 + (c & 3)) * 4;
        else lane_o = (((g >> 1) * 2 * W + (g & 1) + 2 * xlane) * a.out_cs + cg) * 4;
    }
    // Two tiles (accumulator registers r, r + 1 = neighbours in x) per step, in packed fp32: the epilogue's VALU work is not hidden by
    // anything (the wave's MFMAs are over), v_pk_* does two values per instruction in the same order of operations as the scalar form.
    const f32x2 bsbt = {bs, bt}, sl2 = {MODE == 1 ? pre : uslope, MODE == 1 ? pre : uslope};
#pragma unroll
    for (int rp = 0; rp < 8; ++rp) {
        const int r = 2 * rp;
        const int m0 = 8 * (r >> 2) + (r & 3);          // tile index of register r without the lane's half (added through xoff); r + 1: the next tile in x
        const int tyy = m0 / RTX, txx0 = m0 % RTX;      // RTX 8: (r >> 2, r & 3); RTX 16: (r >> 3, 8 * ((r >> 2) & 1) + (r & 3))
        f32x2 s0[4], s1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x2 q0 = {acc[c][r], acc[c][r + 1]}, q1 = {acc[4 + c][r], acc[4 + c][r + 1]}, q2 = {acc[8 + c][r], acc[8 + c][r + 1]},
                        q3 = {acc[12 + c][r], acc[12 + c][r + 1]};
            s0[c] = wino_pk_add(wino_pk_add(q0, q1), q2);
            s1[c] = wino_pk_sub(wino_pk_sub(q1, q2), q3);
        }
        f32x2 y[4];
        y[0] = wino_pk_add(wino_pk_add(s0[0], s0[1]), s0[2]);
        y[1] = wino_pk_sub(wino_pk_sub(s0[1], s0[2]), s0[3]);
        y[2] = wino_pk_add(wino_pk_add(s1[0], s1[1]), s1[2]);
        y[3] = wino_pk_sub(wino_pk_sub(s1[1], s1[2]), s1[3]);
#pragma unroll
        for (int ey = 0; ey < 2; ++ey) {
            const int oy = oy0 + 2 * tyy + ey;          // wave-uniform
            if (FULL || oy < H) {
                const int rowo = SHUF == 1 ? oy * 4 * Ws4 * 16 : (SHUF == 2 ? oy * 4 * W * a.out_cs * 4 : oy * W * a.out_cs * 4), rowr = oy * W * a.res_cs * 4;
#pragma unroll
                for (int ex = 0; ex < 2; ++ex) {
                    f32x2 v2 = wino_pk_mul_hi(wino_pk_add_lo(y[ey * 2 + ex], bsbt), bsbt);
                    f32x2 w2 = v2;
                    if (MODE == 0 || MODE == 1) w2 = wino_pk_mul(v2, sl2);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {       // tile r (tt = 0) and tile r + 1
                        const int tx = txx0 + tt;
                        const int xq = (tx >> 3) * 16 + 2 * (tx & 7) + ex;     // column inside the region = xq + 8 * half (txx = tx + 4 * half)
                        const bool ok = FULL || interior || xlane + xq < W;
                        float v = tt ? v2.y : v2.x;
                        if (MODE == 0) {
                            asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(tt ? w2.y : w2.x));      // (fmaxf on asm results: + 2 canonicalising v_max each)
                        } else if (MODE == 1) {
                            v = v > 0.f ? v : (tt ? w2.y : w2.x);                                          // PReLU with this lane's slope (any sign / size)
                        } else {
                            if (has_res) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, ok ? lane_r : (int)0x80000000, rowr + xq * a.res_cs * 4, 0));
                            if (MODE == 11) v = v > 0.f ? v : v * a.slope;
                            else if (MODE == 12) v = fminf(fmaxf(v, 0.f), 1.f);
                            else if (MODE == 13) v = v > 0.f ? v : v * pre;
                            else if (MODE == 14) v = 1.0f / (1.0f + expf(-v));
                            else if (MODE == 15) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                            v = v * ps + sh;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, ok ? lane_o : (int)0x80000000, rowo + (SHUF == 1 ? xq * 64 : (SHUF == 2 ? 2 * xq * a.out_cs * 4 : xq * a.out_cs * 4)), 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one tile pair at a time: left alone, hipcc hoists all 256 accumulator reads (spills)
    }
}

template <int RTX, int MODE, int SHUF = 0, int PROBE = 0>
__global__ __launch_bounds__(256) void conv_wino_kernel(const WinoArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool EXT = MODE >= 10;     // a general epilogue (wino_epilogue's MODE 10 + act): one kernel per activation, no switch in the item loop
    using G = WinoGeom<RTX>;
    constexpr int PW = G::PW, RW = G::RW, RH = G::RH, NA = G::NA;
    const ConvArgs& a = p.a;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C8 = a.Cin_p >> 3;
    const int H = a.Hin, W = a.Win;
    // clock probe: workgroup 0 stamps shader cycles + the constant-rate counter at its start (drained before the first LDS-DMA piece
    // is issued: the K loop's vmcnt arithmetic counts DMA pieces only) and at its end
    int* const clk_slot = (int*)(smem + G::NBUF * G::BUF_FLOATS + G::TAB_FLOATS + 3 * G::MAXCO);      // the record index waits in LDS
    if (blockIdx.x == 0 && tid == 0) {
        unsigned long long* const cb = g_clk_buf_dev;
        int slot = -1;
        if (cb) {
            const unsigned sl = atomicAdd(&g_clk_next_dev, 1u);
            if (sl < g_clk_cap_dev) {
                slot = (int)sl;
                cb[kClkRecWords * sl + 4] = (unsigned long long)p.clk_tag;
                cb[kClkRecWords * sl + 1] = __builtin_amdgcn_s_memrealtime();
                cb[kClkRecWords * sl + 0] = __builtin_amdgcn_s_memtime();
            }
        }
        *clk_slot = slot;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // Lane-derived address pieces are cheap to recompute and expensive to keep: 256 AGPRs hold the accumulators, the 256 VGPRs are
    // for the patch / operands.  LICM would hoist every lane-only expression out of the loops and then SPILL it (a scratch reload in
    // the hot loop waits vmcnt(0), i.e. for the LDS-DMA in flight): an opaque copy of the lane id per use keeps them local.
    auto opaque_lane = [&]() {
        int l;   // lane id from the exec mask, by volatile asm: the builtin form is CSE'd into ONE value that then lives (and, under pressure,
                 // is spilled) through the whole kernel — its reload at the top of the epilogue sat behind s_waitcnt vmcnt(0)
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    const int img_floats = H * W * a.in_cs;

    // ---- work order -----------------------------------------------------------------------------------------------------
    struct Cur {
        int nb, n, Ry0, Rx0;
        bool valid;    // this wave's region exists
    };
    const int gsz = gridDim.x, wg = blockIdx.x;
    auto item = [&](int i, Cur& c) -> bool {
        int quad;
        if (p.xcd_map) {
            const int x = wg & 7, slot = wg >> 3, S = gsz >> 3;
            const int jl = slot + i * S;
            const int nqx = x < p.NQ ? (p.NQ - x + 7) >> 3 : 0;
            if (jl >= nqx * p.NY) return false;
            const int q = wino_div(jl, p.NY, p.inv_NY);
            quad = x + 8 * q;
            c.nb = jl - q * p.NY;
        } else {
            const int idx = wg + i * gsz;
            if (idx >= p.NQ * p.NY) return false;
            quad = wino_div(idx, p.NY, p.inv_NY);
            c.nb = idx - quad * p.NY;
        }
        const int rg = quad * 4 + wave;
        c.valid = rg < p.R;
        const int rr = c.valid ? rg : 0;
        const int per = p.rx * p.ry;
        c.n = wino_div(rr, per, p.inv_per);
        const int rem = rr - c.n * per;
        const int ryi = wino_div(rem, p.rx, p.inv_rx);
        c.Ry0 = ryi * RH;
        c.Rx0 = (rem - ryi * p.rx) * RW;
        return true;
    };

    // ---- activation DMA: per lane and piece, the (py, px, q) it fetches (kernel constants), then per region the byte offsets
    // The cursor's per-lane byte offsets live in a per-wave LDS table: six more registers held through the loop were six scratch
    // reloads per chunk (each behind s_waitcnt vmcnt(0), i.e. behind the LDS-DMA in flight).
    int* const avtab = (int*)(smem + G::NBUF * G::BUF_FLOATS) + wave * (NA * 64);
    auto make_avoff = [&](const Cur& c) {
        const int ol = opaque_lane();
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int sp = i * 64 + ol;                      // swizzled slot = where the DMA puts this lane's 16 bytes
            const int py = (sp >> 1) / PW;
            const int sl = sp ^ wino_key(py);
            const int code = sp < G::NITEM ? (sl & 1) : -1;
            const int iy = c.Ry0 - 1 + py, ix = c.Rx0 - 1 + ((sl >> 1) - py * PW);
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);       // replicate padding = edge clamp
            const bool ok = code >= 0 && c.valid && (a.pad_replicate || (cy == iy && cx == ix));
            avtab[i * 64 + ol] = ok ? ((cy * W + cx) * a.in_cs + (code & 1) * 4) * 4 : (int)0x80000000;
        }
    };
    // One DMA piece per call: the K loop spreads a chunk's pieces over its MFMA groups (ten in a row back up the texture
    // addresser, which all four waves of the workgroup share, and stall the issuing wave).  At the stream's tail the cursor's
    // descriptors are null (num_records 0: zero fill, no memory traffic): the piece count per chunk — and with it the vmcnt
    // arithmetic of the loop — never changes, and no branch is needed.
    auto issue_a = [&](const __amdgpu_buffer_rsrc_t& rsrc, int i, int voff, int k, int buf) {
        float* abuf = smem + buf * G::BUF_FLOATS + wave * G::A_FLOATS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(abuf + i * 256), 16, voff, k * 32, 0, 0);
    };
    // The lane's byte offset inside a 1 KiB piece / fragment row, HELD in a register through the K loop: recomputed per use it was
    // ~12 isolated VALU instructions per chunk, and an isolated VALU behind an MFMA costs the matrix pipe ~19 cycles
    // (tools/micro/mfma_shadow.hip).  Re-derived from an opaque lane id at every item start, so that it is not live through the
    // epilogue: there the register allocator wants every VGPR for the accumulators' way out of the AGPRs and would spill it — and a
    // scratch reload waits vmcnt(0), i.e. for all LDS-DMA in flight.
    int lane16 = opaque_lane() * 16;
    auto issue_b = [&](const __amdgpu_buffer_rsrc_t& rsrc, int i, int nb, int k, int buf) {
        float* bbuf = smem + buf * G::BUF_FLOATS + 4 * G::A_FLOATS;
        const int piece = wave + 4 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bbuf + piece * 256), 16, lane16, ((nb * C8 + k) * 16 + piece) * 1024, 0, 0);
    };
    auto load_avoff = [&](int(&av)[NA]) {
        const char* const avaddr = (const char*)avtab + (lane16 >> 2);       // this lane's column of the offset table
#pragma unroll
        for (int i = 0; i < NA; ++i) av[i] = *(const int*)(avaddr + i * 256);
    };

    // ---- patch read offsets (floats inside the wave's A image): 4 lane-dependent bases + compile-time (dy, dx) offsets
    f32x4 P[16];
    f32x16 acc[16];
    // patch read addresses (16-byte slots, provably aligned for ds_read_b128): 4 lane-dependent bases (index (dx & 1) * 2 + (dy >> 1),
    // set per chunk by pk_bases below) + compile-time (dy, dx) offsets
    int pb[4];
    auto read_patch_row = [&](int dy) {
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) P[dy * 4 + dx] = ((const f32x4*)smem)[pb[(dx & 1) * 2 + (dy >> 1)] + (((dy * PW + dx) * 2) & ~3)];
    };
    auto read_patch = [&]() {
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) read_patch_row(dy);
    };
    auto read_b = [&](int buf, int j, f32x4(&B)[4]) {
        const char* sB = (const char*)(smem + buf * G::BUF_FLOATS + 4 * G::A_FLOATS) + lane16;
#pragma unroll
        for (int q = 0; q < 4; ++q) B[q] = *(const f32x4*)(sB + (j * 4 + q) * 1024);
    };

    // ---- per-channel epilogue constants in LDS (bias | beta | PReLU slope, padded channels: 0 | 1 | 0).  A global load in the
    // epilogue would be waited for with s_waitcnt vmcnt(0) — behind every LDS-DMA piece in flight, 1-2 us per work item.
    float* const cst = smem + G::NBUF * G::BUF_FLOATS + G::TAB_FLOATS;
    for (int c = tid; c < a.Cout_p; c += 256) {
        const bool real = c < a.Cout;
        cst[c] = real ? a.bias[c] : 0.f;
        cst[G::MAXCO + c] = (real && a.beta) ? a.beta[c] : 1.f;
        cst[2 * G::MAXCO + c] = (real && (EXT || MODE == 1) && a.act == 3) ? a.prelu[c] : 0.f;
    }
    // ---- prologue: three chunks of DMA in flight
    Cur dcur, ccur;
    int d_it = 0, d_k = 0;
    bool d_ok = item(0, dcur);
    if (!d_ok) return;
    make_avoff(dcur);
    auto make_wrsrc = [&](bool live) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, live ? 16 * a.Cin_p * a.Cout_p * 4 : 0, 0x00020000);
    };
    auto make_arsrc = [&](int n, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * img_floats), 0, live ? img_floats * 4 : 0, 0x00020000);
    };
    __amdgpu_buffer_rsrc_t drsrc = make_arsrc(dcur.n, true), dwrsrc = make_wrsrc(true);
    auto dma_issue_all = [&](int buf) {       // prologue: a whole chunk at once
        int av[NA];
        load_avoff(av);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            issue_a(drsrc, i, av[i], d_k, buf);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            issue_b(dwrsrc, i, dcur.nb, d_k, buf);
    };
    auto dma_advance = [&]() {
        if (d_ok && ++d_k == C8) {
            d_k = 0;
            d_ok = item(++d_it, dcur);
            if (d_ok) make_avoff(dcur);
            drsrc = make_arsrc(d_ok ? dcur.n : 0, d_ok);
            dwrsrc = make_wrsrc(d_ok);
            if (!d_ok) dcur.nb = 0;
        }
    };
    // DMA pieces a wave issues per chunk; LDS-DMA completes in order, so "chunk g+1 landed, chunk g+2 may still fly" is vmcnt(NPC)
    constexpr int NPC = NA + 4;
    dma_issue_all(0);
    dma_advance();
    dma_issue_all(1);
    dma_advance();
    dma_issue_all(2);
    dma_advance();
    __builtin_amdgcn_s_waitcnt(wino_waitcnt(2 * NPC, 0));       // chunk 0 has landed (and the constants are written), chunks 1 and 2 stay in flight
    __builtin_amdgcn_s_barrier();

    {
        // ================================================================================================================
        // The K loop with the input transform in PACKED fp32 (v_pk_add_f32 over channel pairs).  In a one-wave-per-SIMD kernel a
        // VALU instruction is not hidden by the wave's own MFMAs: behind an MFMA it costs the matrix pipe ~15 cycles once plus ~4
        // cycles per instruction (tools/micro/mfma_shadow.hip; LDS reads, DMA issue and SALU are free).  So: half the transform
        // instructions (a lane's patch entry is a float4 of 4 channels = two aligned register pairs: channels (0,1) and (2,3) are
        // transformed together, and the MFMA of channel j takes its A operand straight from the pair's half), in four bursts of 16
        // per chunk, no register copies (operands ping-pong by name), lane-only address parts held in registers.
        //   chunk k, sub-step j (16 MFMAs on channel j, A operands VA.x VA.y VB.x VB.y):
        //     0: B fragments of j=1 | pair (2,3) of chunk k -> VB (from P)       | activation pieces 0,1 of chunk k+3
        //     1: wait: own activation pieces of chunk k+1 landed | B of j=2 | patch of chunk k+1 -> P | pieces 2..4
        //     2: B of j=3 | piece 5 (6) | pair (0,1) of chunk k+1 -> VA (from P)
        //     3: wait: own weight pieces of chunk k+1 landed; BARRIER | B of (k+1, j=0) | weight pieces of chunk k+3
        //   LDS buffer k % 3 is the target of chunk k+3: its activation image (wave-private) is free since the patch of chunk k
        //   was read in chunk k-1, its weight tile (shared) once every wave has passed chunk k's barrier with its reads complete.
        //   LDS-DMA completes in order, so "landed" is a compile-time vmcnt: what was issued after the pieces in question.
        f32x2 VA[16], VB[16], t2[16];
        f32x4 Be[4], Bo[4];
        int avp[NA];
        constexpr int NAe = NA, NBe = 4;
        constexpr int W1 = NBe + NAe + NBe + (NAe < 2 ? NAe : 2);      // sub-step 1: newer than the activation pieces of chunk k+1
        constexpr int W3 = NAe + NBe + NAe;                            // sub-step 3: newer than the weight pieces of chunk k+1
        // lane-only parts, once: patch read slots (4), the weight pieces' lane offset, the offset table's lane address
        // (the general-epilogue kernels and the 32x4-region form recompute them per chunk: four more registers live through their epilogue were a scratch
        // reload there — behind s_waitcnt vmcnt(0), i.e. behind all LDS-DMA in flight)
        constexpr bool HOLD = !EXT && RTX == 8;
        int plane[4];
        auto lane_bases = [&](int(&out)[4]) {
            const int ol = opaque_lane();
            const int half = ol >> 5, ty = (ol & 31) / RTX, tx = (ol & 31) % RTX;
            const int base = (2 * ty * PW + 2 * tx) * 2;
#pragma unroll
            for (int dxp = 0; dxp < 2; ++dxp)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) out[dxp * 2 + kk] = wave * (G::A_FLOATS / 4) + base + ((half + 2 * dxp) ^ ((ty + kk) & 3));
        };
        auto pk_bases = [&](int buf) {
            if (!HOLD) {
                lane_bases(pb);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pb[i] += buf * (G::BUF_FLOATS / 4);
                    asm volatile("" : "+v"(pb[i]));
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) pb[i] = plane[i] + buf * (G::BUF_FLOATS / 4);
            }
        };
        auto pk_add = [](f32x2 x, f32x2 y) { return wino_pk_add(x, y); };
        auto pk_sub = [](f32x2 x, f32x2 y) { return wino_pk_sub(x, y); };
        auto tf1 = [&](int hi) {       // rows: t = B^T d, for the channel pair (0,1) (hi = 0) or (2,3) (hi = 1)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x2 d0 = hi ? P[c].hi : P[c].lo, d1 = hi ? P[4 + c].hi : P[4 + c].lo, d2 = hi ? P[8 + c].hi : P[8 + c].lo,
                            d3 = hi ? P[12 + c].hi : P[12 + c].lo;
                t2[c] = pk_sub(d0, d2);
                t2[4 + c] = pk_add(d1, d2);
                t2[8 + c] = pk_sub(d2, d1);
                t2[12 + c] = pk_sub(d1, d3);
            }
        };
        auto tf2 = [&](f32x2(&V)[16]) {       // columns: V = t B
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                V[r * 4 + 0] = pk_sub(t2[r * 4], t2[r * 4 + 2]);
                V[r * 4 + 1] = pk_add(t2[r * 4 + 1], t2[r * 4 + 2]);
                V[r * 4 + 2] = pk_sub(t2[r * 4 + 2], t2[r * 4 + 1]);
                V[r * 4 + 3] = pk_sub(t2[r * 4 + 1], t2[r * 4 + 3]);
            }
        };
#define WINO_MF4(G_, V_, C_, B_)                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                  \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                                    \
        acc[(G_) * 4 + e_] = __builtin_amdgcn_mfma_f32_32x32x2f32(V_[(G_) * 4 + e_].C_, B_[G_][e_], acc[(G_) * 4 + e_], 0, 0, 0);
#define WINO_A(I_) if ((I_) < NA) issue_a(drsrc, (I_), avp[(I_) < NA ? (I_) : 0], d_k, buf)
#define WINO_B(I_) issue_b(dwrsrc, (I_), dcur.nb, d_k, buf)
        // PROBE: sums of stamps (see g_wino_probe_out).  1: q0 / q1 = before / after sub-step 1's vmcnt wait.  2: q0 / q1 = before sub-step 3's
        // wait / after its barrier.  4: q0..q3 = the starts of sub-steps 0..3.  3: q0..q3 = item start, K loop start, K loop end, epilogue end.
        unsigned q0 = 0, q1 = 0, q2 = 0, q3 = 0, qn = 0;
        unsigned long long sa = 0, sb = 0, sc = 0, sd = 0;
#define WINO_STAMP(COND_, V_)                                    \
    if (PROBE != 0 && (COND_)) {                                 \
        __builtin_amdgcn_sched_barrier(0);                       \
        V_ = __builtin_amdgcn_s_memtime();                       \
        __builtin_amdgcn_sched_barrier(0);                       \
    }
        unsigned gchunk = 0;                 // chunk counter of this workgroup's stream: LDS buffer = gchunk % 3
        for (int it = 0; item(it, ccur); ++it) {
            WINO_STAMP(PROBE == 3, sa);
            // An item starts from LDS (nothing but the accumulators crosses the previous item's epilogue): its first chunk landed and
            // became visible in the previous chunk's sub-steps 1 / 3 (or the prologue).
            lane16 = opaque_lane() * 16;
            if (HOLD) lane_bases(plane);
            pk_bases((int)(gchunk % G::NBUF));
            read_patch();
            read_b((int)(gchunk % G::NBUF), 0, Be);
#pragma unroll
            for (int x = 0; x < 16; ++x)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;
            tf1(0);
            tf2(VA);
            WINO_STAMP(PROBE == 3, sb);
            for (int k = 0; k < C8; ++k, ++gchunk) {
                const int buf = (int)(gchunk % G::NBUF), nbuf = buf == G::NBUF - 1 ? 0 : buf + 1;
                // ---- sub-step 0
                WINO_STAMP(PROBE == 4, sa);
                WINO_MF4(0, VA, x, Be);
                read_b(buf, 1, Bo);
                load_avoff(avp);
                WINO_MF4(1, VA, x, Be);
                tf1(1);
                WINO_MF4(2, VA, x, Be);
                tf2(VB);
                WINO_MF4(3, VA, x, Be);
                WINO_A(0); WINO_A(1);
                pk_bases(nbuf);
                // ---- sub-step 1
                __builtin_amdgcn_sched_barrier(0);
                WINO_STAMP(PROBE == 1, sa);
                WINO_STAMP(PROBE == 4, sb);
                __builtin_amdgcn_s_waitcnt(wino_waitcnt(W1, 15));
                WINO_STAMP(PROBE == 1, sb);
                WINO_MF4(0, VA, y, Bo);
                read_b(buf, 2, Be);
                read_patch_row(0);
                WINO_MF4(1, VA, y, Bo);
                read_patch_row(1);
                WINO_A(2);
                WINO_MF4(2, VA, y, Bo);
                read_patch_row(2);
                WINO_A(3);
                WINO_MF4(3, VA, y, Bo);
                read_patch_row(3);
                WINO_A(4);
                // ---- sub-step 2
                WINO_STAMP(PROBE == 4, sc);
                WINO_MF4(0, VB, x, Be);
                read_b(buf, 3, Bo);
                WINO_A(5);
                WINO_MF4(1, VB, x, Be);
                tf1(0);
                WINO_MF4(2, VB, x, Be);
                tf2(VA);
                WINO_MF4(3, VB, x, Be);
                WINO_A(6);
                // ---- sub-step 3
                __builtin_amdgcn_sched_barrier(0);
                if (PROBE == 2) { q1 += (unsigned)sb; }      // the previous chunk's "after the barrier" (0 before the first: sb starts at 0)
                WINO_STAMP(PROBE == 2, sa);
                WINO_STAMP(PROBE == 4, sd);
                __builtin_amdgcn_s_waitcnt(wino_waitcnt(W3, 0));
                __builtin_amdgcn_s_barrier();
                if (PROBE == 1) { q0 += (unsigned)sa; q1 += (unsigned)sb; ++qn; }
                if (PROBE == 2) { q0 += (unsigned)sa; ++qn; }
                if (PROBE == 4) { q0 += (unsigned)sa; q1 += (unsigned)sb; q2 += (unsigned)sc; q3 += (unsigned)sd; ++qn; }
                WINO_STAMP(PROBE == 2, sb);
                WINO_MF4(0, VB, y, Bo);
                read_b(nbuf, 0, Be);
                WINO_B(0);
                WINO_MF4(1, VB, y, Bo);
                WINO_B(1);
                WINO_MF4(2, VB, y, Bo);
                WINO_B(2);
                WINO_MF4(3, VB, y, Bo);
                WINO_B(3);
                __builtin_amdgcn_sched_barrier(0);
                // the next chunk's operands are consumed HERE: LLVM's code sinking otherwise moves their computation below the
                // cursor-advance branch that follows, out from under the MFMAs
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(VA[i]));
                dma_advance();
            }
        // ---- epilogue: Y = A^T M A per tile in registers, + bias, * beta, (+ residual), activation, NHWC store
            WINO_STAMP(PROBE == 3, sc);
            if (ccur.valid) {
                const int ole = opaque_lane();
                const int half = ole >> 5;
                const int co = ccur.nb * 32 + (ole & 31);
                const int coc = co < a.Cout ? co : a.Cout - 1;
                if (ccur.Ry0 + RH <= H && ccur.Rx0 + RW <= W)
                    wino_epilogue<RTX, MODE, true, SHUF>(acc, a, ccur.n, ccur.Ry0, ccur.Rx0, co, coc, half, cst[co], cst[G::MAXCO + co], cst[2 * G::MAXCO + co], p.LO, p.inv_LO);
                else
                    wino_epilogue<RTX, MODE, false, SHUF>(acc, a, ccur.n, ccur.Ry0, ccur.Rx0, co, coc, half, cst[co], cst[G::MAXCO + co], cst[2 * G::MAXCO + co], p.LO, p.inv_LO);
            }
            if (PROBE == 3) {
                WINO_STAMP(true, sd);
                q0 += (unsigned)sa, q1 += (unsigned)sb, q2 += (unsigned)sc, q3 += (unsigned)sd, ++qn;
            }
        }
        if (PROBE != 0 && blockIdx.x == 0) {
            if (PROBE == 2) q1 += (unsigned)sb;      // the last chunk's
            unsigned long long tend = __builtin_amdgcn_s_memtime();
            const int l = opaque_lane();
            if (l < 8) {
                const unsigned v = l == 0 ? q0 : l == 1 ? q1 : l == 2 ? q2 : l == 3 ? q3 : l == 4 ? qn : l == 5 ? (unsigned)tend : l == 6 ? (unsigned)gchunk : (unsigned)PROBE;
                g_wino_probe_out[wave][l] = v;
            }
        }
#undef WINO_STAMP
#undef WINO_MF4
#undef WINO_A
#undef WINO_B
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's dummy pieces write (zeros) into this workgroup's LDS: drain before exit
    if (blockIdx.x == 0) {
        __builtin_amdgcn_s_barrier();      // the workgroup's last wave is the launch's end (workgroup 0 takes items to the last round)
        if (tid == 0 && *clk_slot >= 0) {
            unsigned long long* const cb = g_clk_buf_dev + kClkRecWords * *clk_slot;
            cb[2] = __builtin_amdgcn_s_memtime();
            cb[3] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
// ---- shader-clock probe ----------------------------------------------------------------------------------------------------------
static std::mutex g_clk_mu;
static std::vector<const char*> g_clk_names;
static std::atomic<bool> g_clk_on{false};
static int g_clk_cap_host = 0;      // records the installed buffer holds (guarded by g_clk_mu)
int clock_probe_tag(const char* name) {
    if (!g_clk_on.load(std::memory_order_acquire)) return -1;
    std::lock_guard<std::mutex> lk(g_clk_mu);
    if ((int)g_clk_names.size() >= g_clk_cap_host) return -1;      // the buffer is full: later launches take no record and no name
    g_clk_names.push_back(name);
    return (int)g_clk_names.size() - 1;
}

int wino_probe_read(unsigned* out32) {      // [4 waves][8]: the sums of the LAST probed launch on the current device
    VFI_CHECK_HIP(hipDeviceSynchronize());
    VFI_CHECK_HIP(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_wino_probe_out), 32 * sizeof(unsigned)));
    return 0;
}

int clock_probe_install(unsigned long long* dev_records, int capacity) {      // on the CURRENT device
    const unsigned cap = dev_records ? (unsigned)capacity : 0u, zero = 0u;
    // A probed launch still in flight reads these symbols again at its end (workgroup 0's closing stamp): changing them under it would
    // send that stamp to the new buffer or to nullptr.  hipMemcpyToSymbol does not order with non-blocking streams, so drain the device.
    VFI_CHECK_HIP(hipDeviceSynchronize());
    VFI_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_clk_next_dev), &zero, sizeof(zero)));
    VFI_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_clk_cap_dev), &cap, sizeof(cap)));
    VFI_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_clk_buf_dev), &dev_records, sizeof(dev_records)));
    return 0;
}

// U = G g G^T per (co, ci), laid out [Cout_p/32][Cin_p/8][j][xi/4][half][co%32][xi%4]; physical input channel
// pc = c8 * 8 + half * 4 + j.  chan_map translates logical to physical input channels (concat windows), nullptr = identity.
void pack_wino3x3(const float* w_oihw, int Cout, int Cin, const int* chan_map, int Cin_p, int Cout_p, std::vector<float>& wp) {
    const int C8 = Cin_p / 8;
    wp.assign((size_t)16 * Cin_p * Cout_p, 0.f);
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) {
            const int pc = chan_map ? chan_map[ci] : ci;
            const float* g = w_oihw + ((size_t)co * Cin + ci) * 9;
            double tmp[4][3], U[4][4];
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 3; ++c) tmp[r][c] = Gm[r][0] * g[0 * 3 + c] + Gm[r][1] * g[1 * 3 + c] + Gm[r][2] * g[2 * 3 + c];
            for (int r = 0; r < 4; ++r)
                for (int c = 0; c < 4; ++c) U[r][c] = tmp[r][0] * Gm[c][0] + tmp[r][1] * Gm[c][1] + tmp[r][2] * Gm[c][2];
            const int nb = co / 32, c32 = co % 32, c8 = pc / 8, hf = (pc % 8) / 4, j = pc % 4;
            for (int xi = 0; xi < 16; ++xi) {
                const size_t idx = ((((((size_t)nb * C8 + c8) * 4 + j) * 4 + (xi >> 2)) * 2 + hf) * 32 + c32) * 4 + (xi & 3);
                wp[idx] += (float)U[xi >> 2][xi & 3];
            }
        }
}

// ConvTranspose2d(Cin, LO, 4, 2, 1) as ONE 3x3 stride-1 pad-1 convolution with 4 * LO output channels: output parity (py, px) of
// the transposed convolution reads the 2x2 input neighbourhood rows {y + py - 1, y + py} x columns {x + px - 1, x + px} with kernel
// taps ky = 3 - py - 2a, kx = 3 - px - 2b (pack_deconv4x4) — four of the nine taps of a 3x3 window centred on (y, x).
// w_iohw [Cin][LO][4][4] -> w3 OIHW [4 * LO][Cin][3][3] with channel g * LO + co, g = 2 py + px; bias repeated per parity.
// No multiplication is saved against the four 2x2 parity convolutions (5 of 9 taps are zero and Winograd's 16 products per 2x2
// tile equal the 4 x 4 direct ones) — what is saved is the padding of LO = 24 to a 32-wide MFMA N tile (4 x 24 = 96 = 3 x 32
// exactly) and the layer runs on the tuned Winograd kernel instead of the grouped direct one.
void pack_deconv_as_conv3x3(const float* w_iohw, const float* bias, int Cin, int LO, std::vector<float>& w3, std::vector<float>& b3) {
    w3.assign((size_t)4 * LO * Cin * 9, 0.f);
    b3.assign((size_t)4 * LO, 0.f);
    for (int g = 0; g < 4; ++g) {
        const int py = g >> 1, px = g & 1;
        for (int co = 0; co < LO; ++co) {
            if (bias) b3[(size_t)g * LO + co] = bias[co];
            for (int t = 0; t < 4; ++t) {
                const int a = t >> 1, b = t & 1;
                const int ky = 3 - py - 2 * a, kx = 3 - px - 2 * b;
                for (int ci = 0; ci < Cin; ++ci)
                    w3[(((size_t)(g * LO + co) * Cin + ci) * 3 + py + a) * 3 + px + b] = w_iohw[(((size_t)ci * LO + co) * 4 + ky) * 4 + kx];
            }
        }
    }
}

// 0 = automatic, 1 = direct kernel only, 2 = Winograd wherever the layer shape allows it.  Set through the test hook
// vfi_test_conv_algo only (both forms of one layer object on one input); never read from the environment.
static std::atomic<int> g_wino_mode{0};
int conv_wino_mode(int set) {
    if (set >= 0) g_wino_mode.store(set, std::memory_order_relaxed);
    return g_wino_mode.load(std::memory_order_relaxed);
}

static int wino_cus(int dev) {
    static std::atomic<int> cus_of[kMaxDevices];
    int c = cus_of[dev].load(std::memory_order_relaxed);
    if (!c) {
        hipDeviceProp_t pr;
        if (hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
        c = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
        cus_of[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

// Is the Winograd kernel the better choice for this layer on this launch?  (3x3 stride 1 only; enough work items to fill the
// chip — coarse pyramid levels with long K stay on the direct kernel's split-K path.)
bool conv_wino_eligible(const ConvArgs& a) {
    const int mode = conv_wino_mode(-1);
    if (mode == 1) return false;
    if (a.ntaps != 9 || a.Hout != a.Hin || a.Wout != a.Win || a.in_plane || a.out_mode != 0) return false;
    if (a.Cin_p % 8 || a.Cout_p % 32 || a.Cout_p > 1024) return false;
    if ((long)a.Hin * a.Win * a.in_cs * 4 >= 0x7fffffffL || (long)a.Hin * a.Win * a.out_cs * 4 >= 0x7fffffffL) return false;
    if (a.res && (long)a.Hin * a.Win * a.res_cs * 4 >= 0x7fffffffL) return false;
    if (mode == 2) return true;
    // decided from the IMAGE, never from the launch's batch (the two kernels sum in different orders: a frame's bits must not depend
    // on its launch mates): work items of a nominal two-image launch — what FILM / M2M / IFRNet / GMFSS issue per pair
    const long regions = 2L * cdiv(a.Hin, 8) * cdiv(a.Win, 16);
    const long items = regions / 4 * (a.Cout_p / 32);
    if (items < 192) return false;
    // r5: ... and not a launch of at most two rounds whose last round is nearly empty.  The kernel is persistent, one workgroup per CU:
    // 288 items on 256 CUs (FILM's 256-channel layers at 67x120) are two rounds at 56 % — the direct kernel with split-K is 1.3-1.5x faster
    // there (profiles/r05_film_algo_ab.txt: 1920 -> 256 @67x120 1.88 vs 1.28 ms, 256 -> 256 0.83 vs 0.62), while 576 items (512 channels,
    // 2.25 rounds) stay 1.6x faster on this kernel.
    if (option(kOptWinoQuant)) {
        // against the NOMINAL 256 compute units of the MI355X this library is built for — not the calling thread's current device, not
        // the reserved-CU setting: the two kernels sum in different orders, so the choice (and with it a frame's low-order bits) must be a
        // function of the layer and the image alone, the same on every device, thread and launch
        const long cus = 256;
        const long rounds = (items + cus - 1) / cus;
        if (rounds <= 2 && items * 100 < rounds * cus * 60) return false;
    }
    return true;
}

template <int RTX, int MODE, int SHUF = 0, int PROBE = 0>
static int wino_launch_t(WinoArgs& p, hipStream_t s, const char* name) {
    using G = WinoGeom<RTX>;
    ConvArgs& a = p.a;
    p.rx = cdiv(a.Win, G::RW);
    p.ry = cdiv(a.Hin, G::RH);
    p.R = a.N * p.rx * p.ry;
    p.NQ = cdiv(p.R, 4);
    p.NY = a.Cout_p / 32;
    VFI_REQUIRE((long)p.NQ * p.NY + 2048 < (1L << 24) && p.R < (1 << 24), "conv_wino %s: too many work items for the kernel's float work decode", name);
    p.inv_NY = 1.0f / (float)p.NY, p.inv_per = 1.0f / (float)(p.rx * p.ry), p.inv_rx = 1.0f / (float)p.rx;
    p.LO = a.Cout >= 4 ? a.Cout / 4 : 1;
    p.inv_LO = 1.0f / (float)p.LO;
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "conv_wino %s: device index %d out of range", name, dev);
    static std::atomic<int> attr_set[kMaxDevices];
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        VFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<RTX, MODE, SHUF, PROBE>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        attr_set[dev].store(1, std::memory_order_release);
    }
    const int cus = launch_cus(wino_cus(dev));      // persistent: one workgroup per compute unit it may use
    const long items = (long)p.NQ * p.NY;
    int grid = (int)(items < cus ? items : cus);
    grid = round_up(grid, 8);
    p.xcd_map = option(kOptWinoXcd) ? 1 : 0;
    p.clk_tag = clock_probe_tag(name);
    TraceScope ts(name, s);
    hipLaunchKernelGGL((conv_wino_kernel<RTX, MODE, SHUF, PROBE>), dim3(grid), dim3(256), G::LDS_BYTES, s, p);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// a.w must point at pack_wino3x3's output (device).  variant: 0 = pick, 8 / 16 = region shape (tiles per row)
int conv_wino_launch(const ConvArgs& a, int variant, hipStream_t s, const char* name) {
    VFI_REQUIRE(a.ntaps == 9 && a.Hout == a.Hin && a.Wout == a.Win && !a.in_plane && a.out_mode >= 0 && a.out_mode <= 2,
                "conv_wino %s: 3x3 stride-1 NHWC layers only", name);
    VFI_REQUIRE(a.Cin_p % 8 == 0 && a.Cout_p % 32 == 0 && a.Cout_p <= 1024 && a.in_cs >= a.Cin_p && a.in_cs % 4 == 0, "conv_wino %s: bad channel padding Cin_p=%d Cout_p=%d in_cs=%d",
                name, a.Cin_p, a.Cout_p, a.in_cs);
    VFI_REQUIRE(((uintptr_t)a.in & 15) == 0 && ((uintptr_t)a.w & 15) == 0, "conv_wino %s: unaligned pointers", name);
    VFI_REQUIRE((long)a.Hin * a.Win * a.in_cs * 4 < 0x7fffffffL, "conv_wino %s: image larger than 2 GiB", name);
    VFI_REQUIRE(a.act != 3 || a.prelu, "conv_wino %s: act 3 needs per-channel slopes", name);
    WinoArgs p;
    p.a = a;
    if (variant == 0) {     // region shape by covered-area efficiency: 16x8 pixels unless 32x4 wastes clearly less
        const double area = (double)a.Hin * a.Win;
        const double e8 = area / ((double)cdiv(a.Win, 16) * 16 * cdiv(a.Hin, 8) * 8);
        const double e16 = area / ((double)cdiv(a.Win, 32) * 32 * cdiv(a.Hin, 4) * 4);
        // 32x4 regions pay 7 DMA pieces per chunk instead of 6 and bank-conflicted patch reads: measured on the RIFE trunk (68x120: 6 %
        // better fill, 3 % slower; 34x60: 11 % better fill, 3 % slower) they only win when the fill differs by more than that
        variant = e16 > e8 * 1.15 ? 16 : 8;
    }
    VFI_REQUIRE(variant == 8 || variant == 16, "conv_wino %s: bad variant %d", name, variant);
    VFI_REQUIRE((a.out_mode == 1 || (long)a.Hin * a.Win * a.out_cs * 4 * (a.out_mode == 2 ? 4 : 1) < 0x7fffffffL) && (!a.res || (long)a.Hin * a.Win * a.res_cs * 4 < 0x7fffffffL),
                "conv_wino %s: output / residual image larger than 2 GiB", name);
    // the hot epilogue: no residual, no post affine, none / LeakyReLU with a slope in [0,1]
    // ... or per-channel PReLU (MODE 1)
    const bool ext = a.res != nullptr || a.post_scale != 0.f || !(a.act == 0 || (a.act == 1 && a.slope >= 0.f && a.slope <= 1.f) || a.act == 3);
    VFI_REQUIRE(a.act >= 0 && a.act <= 5, "conv_wino %s: activation code %d", name, a.act);
    const int mode = ext ? 10 + a.act : (a.act == 3 ? 1 : 0);      // wino_epilogue's MODE: one kernel per general activation
    if (a.out_mode == 1) {      // transposed convolution + PixelShuffle as a 3x3 layer (pack_deconv_as_conv3x3): hot epilogue, 16x8 regions
        VFI_REQUIRE(!ext && a.Cout % 4 == 0 && a.act == 0 && !a.beta, "conv_wino %s: the pixel-shuffle output takes the plain epilogue (bias only)", name);
        VFI_REQUIRE((long)(a.out_planes ? a.out_planes : 2) * 16 * a.Hin * a.Win * 16 < 0x7fffffffL, "conv_wino %s: block output larger than 2 GiB", name);
        return wino_launch_t<8, 0, 1>(p, s, name);
    }
    if (a.out_mode == 2) {      // transposed convolution of a layer object as a 3x3 layer: parities interleaved into NHWC at twice the resolution
        // (hot epilogue only: with the general epilogue — per-channel PReLU, IFUNet's decoders — the form measured SLOWER than the grouped
        // direct kernel, 3.2 vs 2.5 ms per IFUNet frame: two spilled registers in an epilogue that already carries the parity addressing)
        VFI_REQUIRE(!a.res && a.Cout % 4 == 0 && (mode == 0 || mode == 1), "conv_wino %s: the transposed-convolution form takes none / LeakyReLU / per-channel PReLU only", name);
        return mode == 1 ? wino_launch_t<8, 1, 2>(p, s, name) : wino_launch_t<8, 0, 2>(p, s, name);
    }
    if (variant == 8 && mode == 0) {      // the hot instantiation's cycle-ledger forms (test option wino_probe; same results, + stamps)
        switch ((int)option(kOptWinoProbe)) {
            case 1: return wino_launch_t<8, 0, 0, 1>(p, s, name);
            case 2: return wino_launch_t<8, 0, 0, 2>(p, s, name);
            case 3: return wino_launch_t<8, 0, 0, 3>(p, s, name);
            case 4: return wino_launch_t<8, 0, 0, 4>(p, s, name);
            default: break;
        }
    }
#define WINO_DISPATCH(R_)                                              \
    switch (mode) {                                                    \
        case 0: return wino_launch_t<R_, 0>(p, s, name);               \
        case 1: return wino_launch_t<R_, 1>(p, s, name);               \
        case 10: return wino_launch_t<R_, 10>(p, s, name);             \
        case 11: return wino_launch_t<R_, 11>(p, s, name);             \
        case 12: return wino_launch_t<R_, 12>(p, s, name);             \
        case 13: return wino_launch_t<R_, 13>(p, s, name);             \
        case 14: return wino_launch_t<R_, 14>(p, s, name);             \
        default: return wino_launch_t<R_, 15>(p, s, name);             \
    }
    if (variant == 16) {
        WINO_DISPATCH(16)
    }
    WINO_DISPATCH(8)
#undef WINO_DISPATCH
}

}  // namespace vfi

using namespace vfi;

extern "C" {

int vfi_clock_probe(void* dev_records, int capacity) {
    VFI_REQUIRE(capacity >= 0 && (dev_records || capacity == 0), "vfi_clock_probe: bad arguments");
    std::lock_guard<std::mutex> lk(g_clk_mu);
    const bool on = dev_records != nullptr && capacity > 0;
    if (clock_probe_install(on ? (unsigned long long*)dev_records : nullptr, on ? capacity : 0) != 0) return -1;
    if (on) g_clk_names.clear();       // (uninstalling keeps the names: they are read after the measured region)
    if (on) g_clk_cap_host = capacity;
    g_clk_on.store(on, std::memory_order_release);
    return 0;
}

int vfi_clock_probe_names(char* buf, int buf_len) {
    std::lock_guard<std::mutex> lk(g_clk_mu);
    std::string out;
    for (const char* n : g_clk_names) {
        out += n ? n : "?";
        out += "\n";
    }
    VFI_REQUIRE(buf && (int)out.size() + 1 <= buf_len, "vfi_clock_probe_names: buffer too small (%d needed)", (int)out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)g_clk_names.size();
}

}  // extern "C"
