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
};

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
    static constexpr int LDS_BYTES = (NBUF * BUF_FLOATS + TAB_FLOATS + 3 * MAXCO) * 4;
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
// MODE 10 + act: the general form (residual, any activation of ConvArgs::act, post affine).
// FULL: the region lies completely inside the image (wave-uniform, true for all but the last row / column of regions): no
// row branches and no per-store column test (they were ~260 of the epilogue's 1350 instructions).
template <int RTX, int MODE, bool FULL = false>
__device__ __forceinline__ void wino_epilogue(const f32x16 (&acc)[16], const ConvArgs& a, int n, int oy0, int ox0, int co, int coc, int half, float bs,
                                              float bt, float pre) {
    const int H = a.Hin, W = a.Win;
    const float uslope = a.act == 1 ? a.slope : 1.0f;
    const float ps = a.post_scale != 0.f ? a.post_scale : 1.0f, sh = a.post_scale != 0.f ? a.post_shift : 0.0f;
    const __amdgpu_buffer_rsrc_t orsrc =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)n * H * W * a.out_cs), 0, H * W * a.out_cs * 4, 0x00020000);
    const bool has_res = MODE >= 10 && a.res != nullptr;
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_res ? a.res + (size_t)n * H * W * a.res_cs : a.out), 0, has_res ? H * W * a.res_cs * 4 : 0, 0x00020000);
    const int xlane = ox0 + 8 * half;          // this lane's first output column; + xq per store
    const bool cok = co < a.Cout;
    const bool interior = FULL || ox0 + 2 * RTX <= W;  // every column of the region is inside the image (wave-uniform)
    // lane part of the byte offsets in a VGPR (0x80000000 = dropped by the descriptor's range check), per-store part scalar
    const int lane_o = cok ? (xlane * a.out_cs + co) * 4 : (int)0x80000000;
    const int lane_r = cok ? (xlane * a.res_cs + co) * 4 : (int)0x80000000;
    // Two tiles (accumulator registers r, r + 1 = neighbours in x) per step, in packed fp32: the epilogue's VALU work is not hidden by
    // anything (the wave's MFMAs are over), v_pk_* does two values per instruction in the same order of operations as the scalar form.
    const f32x2 bsbt = {bs, bt}, sl2 = {uslope, uslope};
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
                const int rowo = oy * W * a.out_cs * 4, rowr = oy * W * a.res_cs * 4;
#pragma unroll
                for (int ex = 0; ex < 2; ++ex) {
                    f32x2 v2 = wino_pk_mul_hi(wino_pk_add_lo(y[ey * 2 + ex], bsbt), bsbt);
                    f32x2 w2 = v2;
                    if (MODE == 0) w2 = wino_pk_mul(v2, sl2);
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {       // tile r (tt = 0) and tile r + 1
                        const int tx = txx0 + tt;
                        const int xq = (tx >> 3) * 16 + 2 * (tx & 7) + ex;     // column inside the region = xq + 8 * half (txx = tx + 4 * half)
                        const bool ok = FULL || interior || xlane + xq < W;
                        float v = tt ? v2.y : v2.x;
                        if (MODE == 0) {
                            asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(tt ? w2.y : w2.x));      // (fmaxf on asm results: + 2 canonicalising v_max each)
                        } else {
                            if (has_res) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, ok ? lane_r : (int)0x80000000, rowr + xq * a.res_cs * 4, 0));
                            if (MODE == 11) v = v > 0.f ? v : v * a.slope;
                            else if (MODE == 12) v = fminf(fmaxf(v, 0.f), 1.f);
                            else if (MODE == 13) v = v > 0.f ? v : v * pre;
                            else if (MODE == 14) v = 1.0f / (1.0f + expf(-v));
                            else if (MODE == 15) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                            v = v * ps + sh;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, ok ? lane_o : (int)0x80000000, rowo + xq * a.out_cs * 4, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one tile pair at a time: left alone, hipcc hoists all 256 accumulator reads (spills)
    }
}

// ABL: compile-time ablations for timing experiments only (results become wrong): 1 no activation DMA, 2 no weight DMA, 4 no epilogue,
// 8 no chunk barrier, 16 no input transform, 32 no B-fragment reads, 64 no patch reads.  The product kernels are ABL = 0.
template <int RTX, int MODE, int ABL = 0>
__global__ __launch_bounds__(256) void conv_wino_kernel(const WinoArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool EXT = MODE != 0;      // a general epilogue (wino_epilogue's MODE 10 + act): one kernel per activation, no switch in the item loop
    using G = WinoGeom<RTX>;
    constexpr int PW = G::PW, RW = G::RW, RH = G::RH, NA = G::NA;
    const ConvArgs& a = p.a;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;   // (prologue only: the loops take the lane id from opaque_lane())
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C8 = a.Cin_p >> 3;
    const int H = a.Hin, W = a.Win;
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
    auto make_rsrc = [&](int n) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * img_floats), 0, img_floats * 4, 0x00020000);
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
        cst[2 * G::MAXCO + c] = (real && EXT && a.act == 3) ? a.prelu[c] : 0.f;
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
            if (!(ABL & 1)) issue_a(drsrc, i, av[i], d_k, buf);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (!(ABL & 2)) issue_b(dwrsrc, i, dcur.nb, d_k, buf);
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
    constexpr int NPC = ((ABL & 1) ? 0 : NA) + ((ABL & 2) ? 0 : 4);
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
        constexpr int NAe = (ABL & 1) ? 0 : NA, NBe = (ABL & 2) ? 0 : 4;
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
#define WINO_A(I_) if (!(ABL & 1) && (I_) < NA) issue_a(drsrc, (I_), avp[(I_) < NA ? (I_) : 0], d_k, buf)
#define WINO_B(I_) if (!(ABL & 2)) issue_b(dwrsrc, (I_), dcur.nb, d_k, buf)
        unsigned gchunk = 0;                 // chunk counter of this workgroup's stream: LDS buffer = gchunk % 3
        for (int it = 0; item(it, ccur); ++it) {
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
            for (int k = 0; k < C8; ++k, ++gchunk) {
                const int buf = (int)(gchunk % G::NBUF), nbuf = buf == G::NBUF - 1 ? 0 : buf + 1;
                // ---- sub-step 0
                WINO_MF4(0, VA, x, Be);
                if (!(ABL & 32)) read_b(buf, 1, Bo);
                load_avoff(avp);
                WINO_MF4(1, VA, x, Be);
                if (!(ABL & 16)) tf1(1);
                WINO_MF4(2, VA, x, Be);
                if (!(ABL & 16)) tf2(VB);
                WINO_MF4(3, VA, x, Be);
                WINO_A(0);
                WINO_A(1);
                pk_bases(nbuf);
                // ---- sub-step 1
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(wino_waitcnt(W1, 15));
                WINO_MF4(0, VA, y, Bo);
                if (!(ABL & 32)) read_b(buf, 2, Be);
                if (!(ABL & 64)) read_patch_row(0);
                WINO_MF4(1, VA, y, Bo);
                if (!(ABL & 64)) read_patch_row(1);
                WINO_A(2);
                WINO_MF4(2, VA, y, Bo);
                if (!(ABL & 64)) read_patch_row(2);
                WINO_A(3);
                WINO_MF4(3, VA, y, Bo);
                if (!(ABL & 64)) read_patch_row(3);
                WINO_A(4);
                // ---- sub-step 2
                WINO_MF4(0, VB, x, Be);
                if (!(ABL & 32)) read_b(buf, 3, Bo);
                WINO_A(5);
                WINO_MF4(1, VB, x, Be);
                if (!(ABL & 16)) tf1(0);
                WINO_MF4(2, VB, x, Be);
                if (!(ABL & 16)) tf2(VA);
                WINO_MF4(3, VB, x, Be);
                WINO_A(6);
                // ---- sub-step 3
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 8)) {
                    __builtin_amdgcn_s_waitcnt(wino_waitcnt(W3, 0));
                    __builtin_amdgcn_s_barrier();
                }
                WINO_MF4(0, VB, y, Bo);
                if (!(ABL & 32)) read_b(nbuf, 0, Be);
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
            if (ccur.valid && !(ABL & 4)) {
                const int ole = opaque_lane();
                const int half = ole >> 5;
                const int co = ccur.nb * 32 + (ole & 31);
                const int coc = co < a.Cout ? co : a.Cout - 1;
                if (ccur.Ry0 + RH <= H && ccur.Rx0 + RW <= W)
                    wino_epilogue<RTX, MODE, true>(acc, a, ccur.n, ccur.Ry0, ccur.Rx0, co, coc, half, cst[co], cst[G::MAXCO + co], cst[2 * G::MAXCO + co]);
                else
                    wino_epilogue<RTX, MODE, false>(acc, a, ccur.n, ccur.Ry0, ccur.Rx0, co, coc, half, cst[co], cst[G::MAXCO + co], cst[2 * G::MAXCO + co]);
            }
        }
#undef WINO_MF4
#undef WINO_A
#undef WINO_B
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's dummy pieces write (zeros) into this workgroup's LDS: drain before exit
#endif
}

// =====================================================================================================================
// TWO WAVES PER SIMD (experimental, opt-in: VFI_WINO_2WAVE=1 / vfi_conv3x3 variant 102; hot epilogue only, Cout_p <= 256).
// docs/design/winograd.md section 6 (b): a wave = 16 tiles (16x4 output pixels) x 32 channels x all 16 transform positions on
// v_mfma_f32_16x16x4_f32 — two 16-channel N blocks x 4 registers x 16 positions = 128 accumulators, so eight waves fit a CU and
// one wave's patch reads, transform and epilogue run under the other's MFMAs.  No software pipelining inside a wave: waves 0-3
// do  barrier -> read + transform(g) -> MFMA(g) [-> epilogue],  waves 4-7  barrier -> MFMA(g-1) [-> epilogue] -> read + transform(g),
// half a chunk out of phase by construction.
//   lane l = (tile m = l % 16 = ty * 8 + tx, K slot kq = l / 16).  The two MFMAs of a chunk take channels {0,2,4,6} and {1,3,5,7}:
//   lane kq feeds channels 2kq and 2kq+1 = 8 adjacent bytes of its pixel in the LDS image (ds_read_b64), transformed as ONE
//   register pair per position (v_pk_add_f32).  The weight pack (pack_wino16) puts the 4 B operands of one position — (MFMA
//   0/1) x (N block 0/1) — into one 16-byte lane item: 16 ds_read_b128 per chunk.
//   LDS: activation ring 3 x 8 waves x 4 KiB (wave-private: 18x6 pixels x 32 B), weight ring 3 x 16 KiB (shared), offset table,
//   constants: 155 KiB.  DMA per iteration g and wave: B(g+1) x 2, then A(g+2) x 4; "chunk g+1 landed" = vmcnt(4).
struct Wino16 {
    static constexpr int PW = 18, PH = 6, RW = 16, RH = 4;
    static constexpr int NITEM = PW * PH * 2;                 // 216 16-byte items per wave and chunk
    static constexpr int NA = 4;                              // DMA pieces (256 slots)
    static constexpr int A_FLOATS = NA * 256;
    static constexpr int B_FLOATS = 4096;
    static constexpr int NBUF = 3;
    static constexpr int OFF_B = NBUF * 8 * A_FLOATS;
    static constexpr int OFF_TAB = OFF_B + NBUF * B_FLOATS;
    static constexpr int TAB = 8 * NA * 64;
    static constexpr int MAXCO = 256;
    static constexpr int OFF_CST = OFF_TAB + TAB;
    static constexpr int LDS_BYTES = (OFF_CST + 3 * MAXCO) * 4;
};

__global__ __launch_bounds__(512) void conv_wino16_kernel(const WinoArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Wino16;
    constexpr int PW = G::PW, NA = G::NA;
    const ConvArgs& a = p.a;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Which two waves share a SIMD is the hardware's choice (measured: NOT wave % 4 — with groups by wave id the two phases did not
    // overlap at all): each wave reads its SIMD id and takes a ticket per SIMD (group = first / second wave there: 0: transform ->
    // MFMA; 1: MFMA(previous chunk) -> transform) and a workgroup-wide ticket for its region slot.
    int* const tick = (int*)(smem + G::OFF_TAB);           // 5 counters, before the offset table is first written
    if (tid < 8) tick[tid] = 0;
    __syncthreads();
    int grp, slot;
    {
        const int simd = (int)(__builtin_amdgcn_s_getreg((2 - 1) << 11 | 4 << 6 | 4)) & 3;       // HW_REG_HW_ID[5:4] = SIMD_ID
        int g0 = 0, s0 = 0;
        if ((tid & 63) == 0) {
            g0 = atomicAdd(&tick[simd], 1);
            s0 = atomicAdd(&tick[4], 1);
        }
        grp = __builtin_amdgcn_readfirstlane(g0) & 1;
        slot = __builtin_amdgcn_readfirstlane(s0) & 7;
    }
    __syncthreads();
    const int abl = p.xcd_map >> 4;              // timing experiments (VFI_WINO16_ABL; results wrong): 1 no patch reads / transform, 2 no MFMAs,
                                                 // 4 no B reads, 8 no epilogue, 16 no DMA, 32 no chunk barrier
    const int C8 = a.Cin_p >> 3;
    const int H = a.Hin, W = a.Win;
    const int img_floats = H * W * a.in_cs;
    auto lane_id = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    struct Cur {
        int nb, n, Ry0, Rx0;
        bool valid;
    };
    const int gsz = gridDim.x, wg = blockIdx.x;
    auto item = [&](int i, Cur& c) -> bool {      // as conv_wino_kernel; waves w and w + 4 take the upper / lower half of 16x8 region w
        int quad;
        if (p.xcd_map & 1) {
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
        const int rg = quad * 4 + (slot & 3);
        c.valid = rg < p.R;
        const int rr = c.valid ? rg : 0;
        const int per = p.rx * p.ry;
        c.n = wino_div(rr, per, p.inv_per);
        const int rem = rr - c.n * per;
        const int ryi = wino_div(rem, p.rx, p.inv_rx);
        c.Ry0 = ryi * 8 + 4 * (slot >> 2);
        c.Rx0 = (rem - ryi * p.rx) * 16;
        return true;
    };
    int* const avtab = (int*)(smem + G::OFF_TAB) + slot * (NA * 64);
    auto make_avoff = [&](const Cur& c) {
        const int ol = lane_id();
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int sp = i * 64 + ol;                      // slot = stored pixel * 2 + channel quad.  A row stores its 9 even columns, then
            const int pix = sp >> 1, q = sp & 1;             // its 9 odd ones: the 8 tiles of a row then read 8 CONSECUTIVE 32-byte pixels
            const int py = pix / PW, pj = pix - py * PW;     // (tiles are 2 pixels apart: interleaved, a read hit every other bank group 4 times)
            const int px = pj < PW / 2 ? 2 * pj : 2 * (pj - PW / 2) + 1;
            const int qs = q ^ ((py >> 1) & 1);              // the stored quad: rows 2,3 swap a pixel's two 16-byte halves (read_patch)
            const int iy = c.Ry0 - 1 + py, ix = c.Rx0 - 1 + px;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            const bool ok = sp < G::NITEM && c.valid && (a.pad_replicate || (cy == iy && cx == ix));
            avtab[i * 64 + ol] = ok ? ((cy * W + cx) * a.in_cs + qs * 4) * 4 : (int)0x80000000;
        }
    };
    auto issue_a = [&](const __amdgpu_buffer_rsrc_t& rsrc, int i, int voff, int k, int buf) {
        float* abuf = smem + (buf * 8 + slot) * G::A_FLOATS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(abuf + i * 256), 16, voff, k * 32, 0, 0);
    };
    int lane16 = lane_id() * 16;
    auto issue_b = [&](const __amdgpu_buffer_rsrc_t& rsrc, int i, int nb, int k, int buf) {
        float* bbuf = smem + G::OFF_B + buf * G::B_FLOATS;
        const int piece = slot + 8 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(bbuf + piece * 256), 16, lane16, ((nb * C8 + k) * 16 + piece) * 1024, 0, 0);
    };
    float* const cst = smem + G::OFF_CST;
    for (int c = tid; c < a.Cout_p; c += 512) {
        const bool real = c < a.Cout;
        cst[c] = real ? a.bias[c] : 0.f;
        cst[G::MAXCO + c] = (real && a.beta) ? a.beta[c] : 1.f;
    }
    // ---- DMA cursor (one chunk per step; null descriptors at the stream's tail keep the piece count constant)
    Cur dcur, ccur;
    int d_it = 0, d_k = 0;
    bool d_ok = item(0, dcur);
    if (!d_ok) return;
    make_avoff(dcur);
    auto make_wrsrc = [&](bool live) { return __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, live ? 16 * a.Cin_p * a.Cout_p * 4 : 0, 0x00020000); };
    auto make_arsrc = [&](int n, bool live) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)n * img_floats), 0, live ? img_floats * 4 : 0, 0x00020000);
    };
    // two cursors: weights run one chunk ahead of the compute, activations two
    __amdgpu_buffer_rsrc_t arsrc = make_arsrc(dcur.n, true);
    Cur bcur = dcur;
    int b_it = 0, b_k = 0;
    bool b_ok = true;
    __amdgpu_buffer_rsrc_t brsrc = make_wrsrc(true);
    auto a_issue = [&](int buf) {
        const int ol = lane_id();
        int av[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) av[i] = avtab[i * 64 + ol];
#pragma unroll
        for (int i = 0; i < NA; ++i) issue_a(arsrc, i, av[i], d_k, buf);
    };
    auto a_advance = [&]() {
        if (d_ok && ++d_k == C8) {
            d_k = 0;
            d_ok = item(++d_it, dcur);
            if (d_ok) make_avoff(dcur);
            arsrc = make_arsrc(d_ok ? dcur.n : 0, d_ok);
        }
    };
    auto b_issue = [&](int buf) {
        issue_b(brsrc, 0, bcur.nb, b_k, buf);
        issue_b(brsrc, 1, bcur.nb, b_k, buf);
    };
    auto b_advance = [&]() {
        if (b_ok && ++b_k == C8) {
            b_k = 0;
            b_ok = item(++b_it, bcur);
            brsrc = make_wrsrc(b_ok);
            if (!b_ok) bcur.nb = 0;
        }
    };
    // prologue: A(0), B(0), A(1): iteration g then issues B(g+1), A(g+2)
    a_issue(0);
    a_advance();
    b_issue(0);
    b_advance();
    a_issue(1);
    a_advance();

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 P[16], V[16];
    f32x4 acc[16][2];
    auto read_patch = [&](int buf) {
        const int ol = lane_id();
        const int m = ol & 15, kq = ol >> 4;
        // Tile rows ty = 0 / 1 read patch rows dy / 2 + dy at the same time, 2 x 576 bytes = half a bank row apart: the same banks.
        // Rows 2,3 are stored with a pixel's two 16-byte halves swapped (make_avoff), so the two tile rows hit different halves:
        // the lane's half is (kq >> 1) ^ ((ty + (dy >> 1)) & 1) — one base per dy >> 1.
        const int ty = m >> 3;
        const int base0 = ((ty * 2) * PW + (m & 7)) * 32 + (kq & 1) * 8 + ((buf * 8 + slot) * G::A_FLOATS) * 4;
        const int baseA = base0 + (((kq >> 1) ^ (ty & 1)) << 4), baseB = base0 + (((kq >> 1) ^ ((ty + 1) & 1)) << 4);
        // ds_read_b64 by hand: hipcc merges neighbouring loads into ds_read2_b64, which the LDS serves in 4 x 16-lane groups on 32 banks
        // at half the rate (MI355X_MICROARCH.md, LDS table) — 2- to 4-way conflicts for this layout (PMC: 200 M conflict cycles of
        // 345 M LDS cycles); plain b64 goes in 2 x 32 lanes on 64 banks, conflict-free here.  The reads are waited for once, below.
        const unsigned lA = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)smem + baseA);
        const unsigned lB = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)smem + baseB);
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
            for (int dx = 0; dx < 4; ++dx)      // column 2 tx + dx is stored at (dx & 1) * 9 + tx + (dx >> 1)
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(P[dy * 4 + dx]) : "v"(dy < 2 ? lA : lB), "n"((dy * PW + (dx & 1) * (PW / 2) + (dx >> 1)) * 32));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto transform = [&]() {       // V = B^T d B on the channel pair, row by row
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            f32x2 t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (r == 0) t[c] = wino_pk_sub(P[c], P[8 + c]);
                if (r == 1) t[c] = wino_pk_add(P[4 + c], P[8 + c]);
                if (r == 2) t[c] = wino_pk_sub(P[8 + c], P[4 + c]);
                if (r == 3) t[c] = wino_pk_sub(P[4 + c], P[12 + c]);
            }
            V[r * 4 + 0] = wino_pk_sub(t[0], t[2]);
            V[r * 4 + 1] = wino_pk_add(t[1], t[2]);
            V[r * 4 + 2] = wino_pk_sub(t[2], t[1]);
            V[r * 4 + 3] = wino_pk_sub(t[1], t[3]);
        }
    };
    // 64 MFMAs, position-major: the 4 B operands of a position are one 16-byte item per lane.  This iteration's six DMA pieces ride
    // behind the four MFMA groups (all eight waves issuing them right behind the barrier backs up the texture addresser and stalls
    // every wave at its first instruction): B(g+1) x 2 | A(g+2) x 2 | x 1 | x 1 — the order the vmcnt arithmetic of head() assumes.
    auto mfma_chunk = [&](int buf, unsigned g, bool dma) {
        const char* sB = (const char*)(smem + G::OFF_B + buf * G::B_FLOATS) + lane16;
        int av[NA];
        if (dma) {
            const int ol = lane_id();
#pragma unroll
            for (int i = 0; i < NA; ++i) av[i] = avtab[i * 64 + ol];
        }
        const int bbuf = (int)((g + 1) % G::NBUF), abuf = (int)((g + 2) % G::NBUF);
        if (abl & 16) dma = false;
        if (abl & 2) return;
        f32x4 Bq[2][4];                  // four positions at a time, the next four requested before this group's MFMAs (pinned: left
                                         // alone hipcc hoists all 16 reads = 64 registers and spills)
#pragma unroll
        for (int e = 0; e < 4; ++e) Bq[0][e] = *(const f32x4*)(sB + e * 1024);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            __builtin_amdgcn_sched_barrier(0);
            if (q4 < 3 && !(abl & 4)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bq[(q4 + 1) & 1][e] = *(const f32x4*)(sB + ((q4 + 1) * 4 + e) * 1024);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int nb2 = 0; nb2 < 2; ++nb2)
                        acc[q4 * 4 + e][nb2] = __builtin_amdgcn_mfma_f32_16x16x4f32(h ? V[q4 * 4 + e].y : V[q4 * 4 + e].x, Bq[q4 & 1][e][h * 2 + nb2],
                                                                                   acc[q4 * 4 + e][nb2], 0, 0, 0);
            if (dma) {
                if (q4 == 0) b_issue(bbuf);
                if (q4 == 1) issue_a(arsrc, 0, av[0], d_k, abuf), issue_a(arsrc, 1, av[1], d_k, abuf);
                if (q4 == 2) issue_a(arsrc, 2, av[2], d_k, abuf);
                if (q4 == 3) issue_a(arsrc, 3, av[3], d_k, abuf);
            }
        }
        if (dma) {
            b_advance();
            a_advance();
        }
    };
    auto epilogue = [&](const Cur& c) {
        if (!c.valid || (abl & 8)) return;
        const int ol = lane_id();
        const int n16 = ol & 15, mb = ol >> 4;
        const float uslope = a.act == 1 ? a.slope : 1.0f;
        const __amdgpu_buffer_rsrc_t orsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)c.n * H * W * a.out_cs), 0, H * W * a.out_cs * 4, 0x00020000);
#pragma unroll
        for (int nb2 = 0; nb2 < 2; ++nb2) {
            const int co = c.nb * 32 + nb2 * 16 + n16;
            const bool cok = co < a.Cout;
            const float bs = cst[co < G::MAXCO ? co : 0], bt = cst[G::MAXCO + (co < G::MAXCO ? co : 0)];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * mb + r;                  // tile: ty = m >> 3, tx = m & 7
                float s0[4], s1[4];
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float q0 = acc[cc][nb2][r], q1 = acc[4 + cc][nb2][r], q2 = acc[8 + cc][nb2][r], q3 = acc[12 + cc][nb2][r];
                    s0[cc] = (q0 + q1) + q2;
                    s1[cc] = (q1 - q2) - q3;
                }
                float y[4];
                y[0] = (s0[0] + s0[1]) + s0[2];
                y[1] = (s0[1] - s0[2]) - s0[3];
                y[2] = (s1[0] + s1[1]) + s1[2];
                y[3] = (s1[1] - s1[2]) - s1[3];
#pragma unroll
                for (int ey = 0; ey < 2; ++ey)
#pragma unroll
                    for (int ex = 0; ex < 2; ++ex) {
                        const int oy = c.Ry0 + 2 * (m >> 3) + ey, ox = c.Rx0 + 2 * (m & 7) + ex;
                        float v = (y[ey * 2 + ex] + bs) * bt;
                        v = fmaxf(v, v * uslope);
                        const bool ok = cok && oy < H && ox < W;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, ok ? ((oy * W + ox) * a.out_cs + co) * 4 : (int)0x80000000, 0, 0);
                    }
            }
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int x = 0; x < 16; ++x)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) acc[x][n2] = f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- the chunk stream: iteration g = chunk g of this workgroup's items, one barrier each.  Two separate loops (one per wave
    // group) with the same barrier count: in one loop hipcc kept V, the patch and both groups' temporaries live together and spilled.
    // barrier: chunk g has landed (all but this wave's newest NA pieces, A(g+1)); this iteration's DMA — B(g+1) into the buffer group 1
    // finished with before the barrier, A(g+2) into this wave's slot of it — follows inside mfma_chunk (or here when there is none)
    auto head = [&](unsigned g, bool dma_now) {
        __builtin_amdgcn_s_waitcnt(wino_waitcnt(NA, 0));
        if (!(abl & 32)) __builtin_amdgcn_s_barrier();
        if (dma_now) {
            b_issue((int)((g + 1) % G::NBUF));
            b_advance();
            a_issue((int)((g + 2) % G::NBUF));
            a_advance();
        }
    };
    zero_acc();
    if (grp == 0) {
        unsigned g = 0;
        for (int it = 0; item(it, ccur); ++it)
            for (int k = 0; k < C8; ++k, ++g) {
                const int buf = (int)(g % G::NBUF);
                head(g, false);
                if (!(abl & 1)) {
                    read_patch(buf);
                    transform();
                }
                mfma_chunk(buf, g, true);
                if (k == C8 - 1) {
                    epilogue(ccur);
                    zero_acc();
                }
            }
    } else {
        unsigned g = 0;
        bool pend = false;
        Cur pcur;
        for (int it = 0; item(it, ccur); ++it)
            for (int k = 0; k < C8; ++k, ++g) {
                const int buf = (int)(g % G::NBUF);
                head(g, !pend);
                if (pend) {
                    mfma_chunk((int)((g + G::NBUF - 1) % G::NBUF), g, true);      // chunk g - 1
                    if (k == 0) {                                         // ... was the last chunk of the previous item
                        epilogue(pcur);
                        zero_acc();
                    }
                }
                if (!(abl & 1)) {
                    read_patch(buf);
                    transform();
                }
                pend = true;
                pcur = ccur;
            }
        if (pend) {
            mfma_chunk((int)((g + G::NBUF - 1) % G::NBUF), g, false);
            epilogue(pcur);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
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

// Pack for conv_wino16_kernel: per (co / 32, ci / 8) 16 pieces of 1 KiB, piece = transform position xi, lane l = (n = l % 16,
// kq = l / 16), the lane's 4 floats = (MFMA h = 0/1) x (N block nb2 = 0/1):  U[xi][ci = c8 * 8 + 2 * kq + h][co = nb * 32 + nb2 * 16 + n].
void pack_wino16(const float* w_oihw, int Cout, int Cin, const int* chan_map, int Cin_p, int Cout_p, std::vector<float>& wp) {
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
            const int nb = co / 32, nb2 = (co % 32) / 16, n = co % 16, c8 = pc / 8, kq = (pc % 8) / 2, h = pc % 2;
            for (int xi = 0; xi < 16; ++xi) {
                const size_t idx = ((((size_t)nb * C8 + c8) * 16 + xi) * 64 + kq * 16 + n) * 4 + h * 2 + nb2;
                wp[idx] += (float)U[xi >> 2][xi & 3];
            }
        }
}

// 0 = automatic (VFI_CONV_WINOGRAD=0 in the environment keeps every 3x3 on the direct kernel), 1 = direct kernel only,
// 2 = Winograd wherever the layer shape allows it (test hook vfi_test_conv_algo: both forms of one layer object on one input)
static std::atomic<int> g_wino_mode{-1};
int conv_wino_mode(int set) {
    if (set >= 0) g_wino_mode.store(set, std::memory_order_relaxed);
    int m = g_wino_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = getenv("VFI_CONV_WINOGRAD");
        m = (e && e[0] == '0') ? 1 : ((e && e[0] == '2') ? 2 : 0);
        g_wino_mode.store(m, std::memory_order_relaxed);
    }
    return m;
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
    const long regions = (long)a.N * cdiv(a.Hin, 8) * cdiv(a.Win, 16);
    return regions / 4 * (a.Cout_p / 32) >= 192;
}

template <int RTX, int MODE, int ABL = 0>
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
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "conv_wino %s: device index %d out of range", name, dev);
    static std::atomic<int> attr_set[kMaxDevices];
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        VFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<RTX, MODE, ABL>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
        attr_set[dev].store(1, std::memory_order_release);
    }
    const int cus = wino_cus(dev);
    const long items = (long)p.NQ * p.NY;
    int grid = (int)(items < cus ? items : cus);
    grid = round_up(grid, 8);
    static const int xcd = [] { const char* e = getenv("VFI_WINO_XCD"); return (e && e[0] == '0') ? 0 : 1; }();     // A/B hook
    p.xcd_map = xcd;
    TraceScope ts(name, s);
    hipLaunchKernelGGL((conv_wino_kernel<RTX, MODE, ABL>), dim3(grid), dim3(256), G::LDS_BYTES, s, p);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// a.w must point at pack_wino16's output (device).  Hot epilogue only (no residual, none / LeakyReLU), Cout_p <= 256.
bool conv_wino16_eligible(const ConvArgs& a, bool any_size) {
    const bool ext = a.res != nullptr || a.post_scale != 0.f || !(a.act == 0 || (a.act == 1 && a.slope >= 0.f && a.slope <= 1.f));
    if (ext || a.Cout_p > Wino16::MAXCO) return false;
    if (!any_size) return conv_wino_eligible(a);
    return a.ntaps == 9 && a.Hout == a.Hin && a.Wout == a.Win && !a.in_plane && a.out_mode == 0 && a.Cin_p % 8 == 0 && a.Cout_p % 32 == 0 &&
           (long)a.Hin * a.Win * a.in_cs * 4 < 0x7fffffffL && (long)a.Hin * a.Win * a.out_cs * 4 < 0x7fffffffL;
}
int conv_wino16_launch(const ConvArgs& a, hipStream_t s, const char* name) {
    VFI_REQUIRE(conv_wino16_eligible(a, true), "conv_wino16 %s: layer not eligible", name);
    VFI_REQUIRE(a.Cin_p % 8 == 0 && a.Cout_p % 32 == 0 && a.in_cs >= a.Cin_p && a.in_cs % 4 == 0 && ((uintptr_t)a.in & 15) == 0 && ((uintptr_t)a.w & 15) == 0,
                "conv_wino16 %s: bad channel padding / alignment", name);
    VFI_REQUIRE((long)a.Hin * a.Win * a.in_cs * 4 < 0x7fffffffL && (long)a.Hin * a.Win * a.out_cs * 4 < 0x7fffffffL, "conv_wino16 %s: image larger than 2 GiB", name);
    WinoArgs p;
    p.a = a;
    p.rx = cdiv(a.Win, 16), p.ry = cdiv(a.Hin, 8);
    p.R = a.N * p.rx * p.ry;
    p.NQ = cdiv(p.R, 4);
    p.NY = a.Cout_p / 32;
    VFI_REQUIRE((long)p.NQ * p.NY + 2048 < (1L << 24) && p.R < (1 << 24), "conv_wino16 %s: too many work items", name);
    p.inv_NY = 1.0f / (float)p.NY, p.inv_per = 1.0f / (float)(p.rx * p.ry), p.inv_rx = 1.0f / (float)p.rx;
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "conv_wino16 %s: device index %d out of range", name, dev);
    static std::atomic<int> attr_set[kMaxDevices];
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        VFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, Wino16::LDS_BYTES));
        attr_set[dev].store(1, std::memory_order_release);
    }
    const int cus = wino_cus(dev);
    const long items = (long)p.NQ * p.NY;
    int grid = (int)(items < cus ? items : cus);
    grid = round_up(grid, 8);
    static const int abl16 = [] { const char* e = getenv("VFI_WINO16_ABL"); return e ? atoi(e) : 0; }();
    p.xcd_map = 1 | (abl16 << 4);
    TraceScope ts(name, s);
    hipLaunchKernelGGL(conv_wino16_kernel, dim3(grid), dim3(512), Wino16::LDS_BYTES, s, p);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// a.w must point at pack_wino3x3's output (device).  variant: 0 = pick, 8 / 16 = region shape (tiles per row)
int conv_wino_launch(const ConvArgs& a, int variant, hipStream_t s, const char* name) {
    VFI_REQUIRE(a.ntaps == 9 && a.Hout == a.Hin && a.Wout == a.Win && !a.in_plane && a.out_mode == 0,
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
    VFI_REQUIRE((long)a.Hin * a.Win * a.out_cs * 4 < 0x7fffffffL && (!a.res || (long)a.Hin * a.Win * a.res_cs * 4 < 0x7fffffffL),
                "conv_wino %s: output / residual image larger than 2 GiB", name);
    // the hot epilogue: no residual, no post affine, none / LeakyReLU with a slope in [0,1]
    const bool ext = a.res != nullptr || a.post_scale != 0.f || !(a.act == 0 || (a.act == 1 && a.slope >= 0.f && a.slope <= 1.f));
    VFI_REQUIRE(a.act >= 0 && a.act <= 5, "conv_wino %s: activation code %d", name, a.act);
    const int mode = ext ? 10 + a.act : 0;      // wino_epilogue's MODE: one kernel per general activation
    if (!ext && variant == 8) {     // A/B hook: VFI_WINO_ABLATE selects a compile-time variant of the hot kernel (see ABL above)
        static const int abl = [] { const char* e = getenv("VFI_WINO_ABLATE"); return e ? atoi(e) : 0; }();
        if (abl == 4) return wino_launch_t<8, 0, 4>(p, s, name);
    }
#define WINO_DISPATCH(R_)                                              \
    switch (mode) {                                                    \
        case 0: return wino_launch_t<R_, 0>(p, s, name);               \
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
