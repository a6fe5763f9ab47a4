// HBM-bound kernels of the RIFE 4.7 hot loop (everything that is not a trunk convolution).
//
// Layouts (all fp32, NHWC):
// "planar4" = a C-channel image stored as C/4 planes of [H][W] float4: every load/store of a wave is a run of
// consecutive 16-byte items (the rocprofv3 TCP/SQ counters of the first, interleaved version showed these
// kernels bound by L1 line touches per instruction, not by HBM: profiles/r01_pmc2_*).
//   frame pack  P[slot] : planar4 [2][Hp][Wp][4]   plane 0 = clamp(rgb,0,1),0 zero-padded to x64,
//                                                   plane 1 = encode(img)
//   flow        F       : [B][Hp][Wp][4] (F01 -> frame0, F23 -> frame1), mask M : [B][Hp][Wp]
//   stage input X       : planar4 [B][CX/4][Hs][Ws][4], channel order of the reference's torch.cat
//   block output T      : planar4 [B][2][Hs][Ws][4]: the PixelShuffle'd lastconv output, plane 0 = flow
//                         delta (4 ch), plane 1 = (mask, unused, 0, 0); written by the conv epilogue
//
// Reference semantics restated (vfi_models/rife/rife_arch.py): warp :31-70; IFBlock's
// F.interpolate calls :238-248,263-266; input torch.cat :543-548,629-644; flow/mask update
// :645,698-699; final blend :721-723,732; encode :414-416.
#include "rife_ops.h"

#include <algorithm>
#include <type_traits>
#include <atomic>
#include "rife_warp.h"
#include <cstdlib>

namespace vfi {

__device__ static inline float4 lerp4(const float4 a, const float4 b, const float4 c, const float4 d,
                                      const Tap4& t) {
    float4 r;
    r.x = a.x * t.nw + b.x * t.ne + c.x * t.sw + d.x * t.se;
    r.y = a.y * t.nw + b.y * t.ne + c.y * t.sw + d.y * t.se;
    r.z = a.z * t.nw + b.z * t.ne + c.z * t.sw + d.z * t.se;
    r.w = a.w * t.nw + b.w * t.ne + c.w * t.sw + d.w * t.se;
    return r;
}

// Sample the frame pack: lo = plane 0 (rgb,0), hi[j] = feature plane 1+j (plane stride hi_off floats).
template <int NHI>
__device__ static inline void sample_pack(const float* __restrict__ P, size_t hi_off, const Tap4& t, float4& lo,
                                          float4* hi) {
    const float4* q = (const float4*)P;
    lo = lerp4(q[t.o00], q[t.o01], q[t.o10], q[t.o11], t);
#pragma unroll
    for (int j = 0; j < NHI; ++j) {
        const float4* h = (const float4*)(P + (size_t)(1 + j) * hi_off);
        hi[j] = lerp4(h[t.o00], h[t.o01], h[t.o10], h[t.o11], t);
    }
}
// channel order of the reference's torch.cat: (img0 3, img1 3, f0 4*NF, f1 4*NF, timestep[, mask, flow 4])
template <int NF>
__device__ static inline void cat_inputs(float* o, const float4& a_lo, const float4& b_lo, const float4* a_hi, const float4* b_hi,
                                         float tstep) {
    o[0] = a_lo.x; o[1] = a_lo.y; o[2] = a_lo.z;
    o[3] = b_lo.x; o[4] = b_lo.y; o[5] = b_lo.z;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        o[6 + 4 * j] = a_hi[j].x; o[7 + 4 * j] = a_hi[j].y; o[8 + 4 * j] = a_hi[j].z; o[9 + 4 * j] = a_hi[j].w;
        o[6 + 4 * NF + 4 * j] = b_hi[j].x; o[7 + 4 * NF + 4 * j] = b_hi[j].y;
        o[8 + 4 * NF + 4 * j] = b_hi[j].z; o[9 + 4 * NF + 4 * j] = b_hi[j].w;
    }
    o[6 + 8 * NF] = tstep;
}

// generic NHWC warp (C arbitrary) — parity-test entry point and building block for other nodes
__global__ void warp_border_kernel(const float* __restrict__ in, const float* __restrict__ flow,
                                   float* __restrict__ out, int N, int H, int W, int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * H * W) return;
    const int X = idx % W;
    const int Y = (idx / W) % H;
    const int n = idx / ((long)W * H);
    const WarpGeo g = make_warp_geo(W, H);
    const float2 f = ((const float2*)flow)[idx];
    const Tap4 t = warp_taps(g, X, Y, f.x, f.y);
    const float* base = in + (size_t)n * H * W * C;
    float* o = out + (size_t)idx * C;
    for (int c = 0; c < C; ++c)
        o[c] = base[(size_t)t.o00 * C + c] * t.nw + base[(size_t)t.o01 * C + c] * t.ne +
               base[(size_t)t.o10 * C + c] * t.sw + base[(size_t)t.o11 * C + c] * t.se;
}

int warp_border_launch(const float* in, const float* flow, float* out, int N, int H, int W, int C,
                       hipStream_t s) {
    const long total = (long)N * H * W;
    TraceScope ts("warp_border", s);
    hipLaunchKernelGGL(warp_border_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, flow,
                       out, N, H, W, C);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// frame preparation: clamp, drop alpha, zero-pad to x64            rife_arch.py:476-484
// ---------------------------------------------------------------------------------------
template <typename SRC>   // float: IMAGE as ComfyUI hands it over; unsigned char: 8-bit frames converted here (x / 255)
__global__ void prep_frame_kernel(const SRC* __restrict__ src, float* __restrict__ P, int H, int W, int C,
                                  int Hp, int Wp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hp * Wp) return;
    const int X = idx % Wp, Y = idx / Wp;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if (Y < H && X < W) {
        const SRC* s = src + ((size_t)Y * W + X) * C;
        if (sizeof(SRC) == 1) {     // torch's uint8 -> float32 image: x.float() / 255 (IEEE division, already in [0, 1])
            v.x = __fdiv_rn((float)s[0], 255.0f);
            v.y = __fdiv_rn((float)s[1], 255.0f);
            v.z = __fdiv_rn((float)s[2], 255.0f);
        } else {
            v.x = fminf(fmaxf((float)s[0], 0.f), 1.f);
            v.y = fminf(fmaxf((float)s[1], 0.f), 1.f);
            v.z = fminf(fmaxf((float)s[2], 0.f), 1.f);
        }
    }
    *(float4*)(P + (size_t)idx * 4) = v;
}

// float32 frames in [0,1] -> 8 bit: round(clamp(x) * 255), ties to even like torch.round; 4 values per thread
__global__ void f32_to_u8_kernel(const float* __restrict__ in, unsigned char* __restrict__ out, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *(const float4*)(in + i);
        uchar4 o;
        o.x = (unsigned char)rintf(fminf(fmaxf(v.x, 0.f), 1.f) * 255.0f);
        o.y = (unsigned char)rintf(fminf(fmaxf(v.y, 0.f), 1.f) * 255.0f);
        o.z = (unsigned char)rintf(fminf(fmaxf(v.z, 0.f), 1.f) * 255.0f);
        o.w = (unsigned char)rintf(fminf(fmaxf(v.w, 0.f), 1.f) * 255.0f);
        *(uchar4*)(out + i) = o;
    } else {
        for (long j = i; j < n; ++j) out[j] = (unsigned char)rintf(fminf(fmaxf(in[j], 0.f), 1.f) * 255.0f);
    }
}

// Head, first layer: Conv2d(3,CM,3,stride 2,pad 1) (+ LeakyReLU(0.2) for the Head / Head_417 variants).
// weights [tap][ci][co] (uniform -> SGPRs).  4.7: encode.0 (CM 16, no activation) rife_arch.py:414-416;
// 4.17: Head_417.cnn0 (CM 32) :355-375; 4.26: Head.cnn0 (CM 16) :378-398.
template <int CM, bool ACT>
__global__ void encode_conv_kernel(const float* __restrict__ P, const float* __restrict__ w,
                                   const float* __restrict__ bias, float* __restrict__ E, int Hp, int Wp) {
    const int He = Hp / 2, We = Wp / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= He * We) return;
    const int x = idx % We, y = idx / We;
    float acc[CM];
#pragma unroll
    for (int co = 0; co < CM; ++co) acc[co] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = 2 * y - 1 + ky;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = 2 * x - 1 + kx;
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < Hp && ix >= 0 && ix < Wp) v = *(const float4*)(P + ((size_t)iy * Wp + ix) * 4);
            const float* wt = w + (ky * 3 + kx) * 3 * CM;
#pragma unroll
            for (int co = 0; co < CM; ++co)
                acc[co] = fmaf(v.z, wt[2 * CM + co], fmaf(v.y, wt[CM + co], fmaf(v.x, wt[co], acc[co])));
        }
    }
    float4* o = (float4*)(E + (size_t)idx * CM);
#pragma unroll
    for (int q = 0; q < CM / 4; ++q) {
        float4 r = make_float4(acc[4 * q] + bias[4 * q], acc[4 * q + 1] + bias[4 * q + 1], acc[4 * q + 2] + bias[4 * q + 2],
                               acc[4 * q + 3] + bias[4 * q + 3]);
        if (ACT) {
            r.x = r.x > 0.f ? r.x : r.x * 0.2f;
            r.y = r.y > 0.f ? r.y : r.y * 0.2f;
            r.z = r.z > 0.f ? r.z : r.z * 0.2f;
            r.w = r.w > 0.f ? r.w : r.w * 0.2f;
        }
        o[q] = r;
    }
}

// Head, last layer: ConvTranspose2d(CM,CF,4,stride 2,pad 1), no activation; one thread = one E pixel = a 2x2
// output quad (all 4 parities).  weights [ky][kx][ci][co(CF)]; result goes to pack planes 1..CF/4.
template <int CM, int CF>
__global__ void encode_deconv_kernel(const float* __restrict__ E, const float* __restrict__ w,
                                     const float* __restrict__ bias, float* __restrict__ P, int Hp, int Wp) {
    const int He = Hp / 2, We = Wp / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= He * We) return;
    const int x = idx % We, y = idx / We;
    float acc[4][CF];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int co = 0; co < CF; ++co) acc[g][co] = bias[co];
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int iy = y + dy;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int ix = x + dx;
            const bool ok = iy >= 0 && iy < He && ix >= 0 && ix < We;
            const float4* e = (const float4*)(E + ((size_t)(ok ? iy : y) * We + (ok ? ix : x)) * CM);
#pragma unroll
            for (int q = 0; q < CM / 4; ++q) {   // 4 input channels at a time: keeps the live set small for CM = 32
                float4 t = e[q];
                if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
                const float ev[4] = {t.x, t.y, t.z, t.w};
                // parity (py,px) uses input offset dy in {py-1, py} with ky = py + 1 - 2*dy
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    if (dy < py - 1 || dy > py) continue;
                    const int ky = py + 1 - 2 * dy;
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        if (dx < px - 1 || dx > px) continue;
                        const int kx = px + 1 - 2 * dx;
                        const float* wt = w + ((ky * 4 + kx) * CM + 4 * q) * CF;
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                            for (int co = 0; co < CF; ++co)
                                acc[py * 2 + px][co] = fmaf(ev[ci], wt[ci * CF + co], acc[py * 2 + px][co]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int Y = 2 * y + (g >> 1), X = 2 * x + (g & 1);
#pragma unroll
        for (int q = 0; q < CF / 4; ++q)
            *(float4*)(P + (size_t)(1 + q) * Hp * Wp * 4 + ((size_t)Y * Wp + X) * 4) =
                make_float4(acc[g][4 * q], acc[g][4 * q + 1], acc[g][4 * q + 2], acc[g][4 * q + 3]);
    }
}

// ---------------------------------------------------------------------------------------
// arch 4.7 frame pack in ONE launch: clamp / pad + encode.0 (Conv 3->16, 3x3 s2) + encode.1 (ConvTranspose 16->4, 4x4 s2)
// ---------------------------------------------------------------------------------------
// The three kernels above move the frame through HBM three times (plane 0 out and back in, the 16-channel half-resolution tensor E
// out and back in: 75 us per 1080p frame, 2.45 ms per 33-frame bench step).  Here a workgroup owns a 32x32 tile of the pack:
//   phase 0: its 37x37 source pixels (the tile, + the halo the two convolutions reach) -> LDS, clamped, zero outside the frame;
//            plane 0 of the tile is written from there;
//   phase 1: the 18x18 E pixels the transposed convolution needs (16 channels) -> LDS, zero outside the half-resolution image
//            (that is the transposed convolution's padding, not conv(zeros) + bias);
//   phase 2: one thread per E pixel of the 16x16 centre = one 2x2 output quad, plane 1 written.
// Same fmaf order as encode_conv_kernel / encode_deconv_kernel: the pack is bit-identical to the three-kernel path
// (tests/test_gpu_rife.py::test_fused_frame_pack_is_bit_identical).  E never exists in HBM; HBM traffic 25 + 67 MB per frame.
constexpr int ENC_T = 32;                      // output tile
constexpr int ENC_TE = ENC_T / 2 + 2;          // E tile (18)
constexpr int ENC_TI = 2 * ENC_TE + 1;         // source tile (37)

template <typename SRC>
__global__ __launch_bounds__(256) void encode47_fused_kernel(const SRC* __restrict__ src, float* __restrict__ P, const float* __restrict__ w0,
                                                             const float* __restrict__ b0, const float* __restrict__ w1, const float* __restrict__ b1,
                                                             int H, int W, int C, int Hp, int Wp, int tiles_x) {
    constexpr int CM = 16, CF = 4;
    __shared__ __attribute__((aligned(16))) float4 sI[ENC_TI * ENC_TI];
    __shared__ __attribute__((aligned(16))) float sE[ENC_TE * ENC_TE * CM];
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int Y0 = ty * ENC_T, X0 = tx * ENC_T;            // first output pixel of the tile
    const int ey0 = Y0 / 2 - 1, ex0 = X0 / 2 - 1;          // first E pixel of the E tile
    const int iy0 = 2 * ey0 - 1, ix0 = 2 * ex0 - 1;        // first source pixel of the source tile
    const int He = Hp / 2, We = Wp / 2;
    // ---- phase 0
    for (int i = tid; i < ENC_TI * ENC_TI; i += 256) {
        const int py = i / ENC_TI, px = i - py * ENC_TI;
        const int Y = iy0 + py, X = ix0 + px;
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if (Y >= 0 && Y < H && X >= 0 && X < W) {
            const SRC* sp = src + ((size_t)Y * W + X) * C;
            if (sizeof(SRC) == 1) {
                v.x = __fdiv_rn((float)sp[0], 255.0f);
                v.y = __fdiv_rn((float)sp[1], 255.0f);
                v.z = __fdiv_rn((float)sp[2], 255.0f);
            } else {
                v.x = fminf(fmaxf((float)sp[0], 0.f), 1.f);
                v.y = fminf(fmaxf((float)sp[1], 0.f), 1.f);
                v.z = fminf(fmaxf((float)sp[2], 0.f), 1.f);
            }
        }
        sI[i] = v;
        // plane 0 of this tile (source-tile offset 3: iy0 = Y0 - 3)
        if (py >= 3 && py < 3 + ENC_T && px >= 3 && px < 3 + ENC_T && Y < Hp && X < Wp) *(float4*)(P + ((size_t)Y * Wp + X) * 4) = v;
    }
    __syncthreads();
    // ---- phase 1: E tile
    for (int i = tid; i < ENC_TE * ENC_TE; i += 256) {
        const int py = i / ENC_TE, px = i - py * ENC_TE;
        const int ey = ey0 + py, ex = ex0 + px;
        float acc[CM];
#pragma unroll
        for (int co = 0; co < CM; ++co) acc[co] = 0.f;
        const bool in_img = ey >= 0 && ey < He && ex >= 0 && ex < We;
        if (in_img) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 v = sI[(2 * py + ky) * ENC_TI + 2 * px + kx];      // source pixel (2 ey - 1 + ky, 2 ex - 1 + kx)
                    const float* wt = w0 + (ky * 3 + kx) * 3 * CM;
#pragma unroll
                    for (int co = 0; co < CM; ++co) acc[co] = fmaf(v.z, wt[2 * CM + co], fmaf(v.y, wt[CM + co], fmaf(v.x, wt[co], acc[co])));
                }
#pragma unroll
            for (int co = 0; co < CM; ++co) acc[co] += b0[co];
        }
#pragma unroll
        for (int q = 0; q < CM / 4; ++q) *(float4*)&sE[i * CM + 4 * q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    __syncthreads();
    // ---- phase 2: one E pixel of the 16x16 centre per thread -> its 2x2 output quad
    {
        const int lx = tid & 15, ly = tid >> 4;
        const int y = Y0 / 2 + ly, x = X0 / 2 + lx;      // E pixel
        if (y >= He || x >= We) return;
        float acc[4][CF];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int co = 0; co < CF; ++co) acc[g][co] = b1[co];
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const float* e = &sE[((ly + 1 + dy) * ENC_TE + lx + 1 + dx) * CM];
#pragma unroll
                for (int q = 0; q < CM / 4; ++q) {
                    const float4 t = *(const float4*)(e + 4 * q);
                    const float ev[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        if (dy < py - 1 || dy > py) continue;
                        const int ky = py + 1 - 2 * dy;
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            if (dx < px - 1 || dx > px) continue;
                            const int kx = px + 1 - 2 * dx;
                            const float* wt = w1 + ((ky * 4 + kx) * CM + 4 * q) * CF;
#pragma unroll
                            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                                for (int co = 0; co < CF; ++co) acc[py * 2 + px][co] = fmaf(ev[ci], wt[ci * CF + co], acc[py * 2 + px][co]);
                        }
                    }
                }
            }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int Y = 2 * y + (g >> 1), X = 2 * x + (g & 1);
            *(float4*)(P + (size_t)Hp * Wp * 4 + ((size_t)Y * Wp + X) * 4) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
        }
    }
}

int encode47_fused_launch(const float* f32, const unsigned char* u8, float* P, const float* w0, const float* b0, const float* w1, const float* b1,
                          int H, int W, int C, int Hp, int Wp, hipStream_t s) {
    const int tiles_x = cdiv(Wp, ENC_T), tiles_y = cdiv(Hp, ENC_T);
    TraceScope ts("encode_fused", s);
    if (u8)
        hipLaunchKernelGGL(encode47_fused_kernel<unsigned char>, dim3(tiles_x * tiles_y), dim3(256), 0, s, u8, P, w0, b0, w1, b1, H, W, C, Hp, Wp, tiles_x);
    else
        hipLaunchKernelGGL(encode47_fused_kernel<float>, dim3(tiles_x * tiles_y), dim3(256), 0, s, f32, P, w0, b0, w1, b1, H, W, C, Hp, Wp, tiles_x);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// The same frame pack for a BATCH of frames in one launch (vfi_rife_load_frames): persistent workgroups, source tile of the NEXT
// work item prefetched into registers while this one computes.
// ---------------------------------------------------------------------------------------
// Why: encode47_fused_kernel's 42.6 KB of LDS admit 3 workgroups per CU, and each of them alternates load -> compute -> compute:
// on average ONE workgroup per CU has loads in flight (16 KB), where ~50 KB per CU are needed to cover the HBM latency at full
// rate (measured r3: 51 us per 1080p frame = 1.8 TB/s for 92 MB; VALU floor 13 us, HBM floor 15 us).  Here every workgroup issues
// the global loads of its next tile right after it has stored the current one into LDS, so they fly under phases 1 and 2
// (~3 us of VALU work), and one launch covers all frames of a step (no per-frame launch tail: 2040 workgroups = 2.7 rounds of
// 768 slots per frame before).  Arithmetic is encode47_fused_kernel's, expression for expression: bit-identical packs
// (tests/test_gpu_rife.py::test_batched_frame_pack_is_bit_identical).
constexpr int ENC_MAXF = 64;                   // frames per launch (kernel-argument arrays)
struct EncodeBatch {
    const void* src[ENC_MAXF];
    float* P[ENC_MAXF];
};
constexpr int ENC_NPRE = (ENC_TI * ENC_TI + 255) / 256;      // source pixels per thread and tile (6)

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is fence + s_barrier and the fence waits for EVERY outstanding
// memory operation of the wave (s_waitcnt vmcnt(0)): the global loads of the next tile, issued just before, would be waited for at
// the very next barrier — no prefetch at all.  Here only the LDS counter is drained; the prefetched registers are waited for by
// hipcc's own wait-count pass where they are first used (the next iteration's phase 0).
__device__ __forceinline__ void enc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename SRC>
__global__ __launch_bounds__(256) void encode47_batch_kernel(const EncodeBatch fb, const float* __restrict__ w0, const float* __restrict__ b0,
                                                             const float* __restrict__ w1, const float* __restrict__ b1, int H, int W, int C, int Hp,
                                                             int Wp, int tiles_x, int tiles_per_frame, int n_items) {
    constexpr int CM = 16, CF = 4;
    __shared__ __attribute__((aligned(16))) float4 sI[ENC_TI * ENC_TI];
    __shared__ __attribute__((aligned(16))) float sE[ENC_TE * ENC_TE * CM];
    const int tid = threadIdx.x;
    const int He = Hp / 2, We = Wp / 2;
    // raw source values of the next item (clamp / divide happen when they are stored to LDS: same values as the single-frame kernel),
    // one 32-bit register each
    typedef typename std::conditional<sizeof(SRC) == 1, unsigned, float>::type RAW;
    RAW pre[ENC_NPRE][3];
    auto decode = [&](int item, int& f, int& Y0, int& X0) {
        f = item / tiles_per_frame;
        const int t = item - f * tiles_per_frame;
        const int ty = t / tiles_x;
        Y0 = ty * ENC_T, X0 = (t - ty * tiles_x) * ENC_T;
    };
    auto prefetch = [&](int item) {
        int f, Y0, X0;
        decode(item, f, Y0, X0);
        const SRC* src = (const SRC*)fb.src[f];
        const int iy0 = Y0 - 3, ix0 = X0 - 3;
#pragma unroll
        for (int r = 0; r < ENC_NPRE; ++r) {
            const int i = tid + r * 256;
            const int py = i / ENC_TI, px = i - py * ENC_TI;
            // UNCONDITIONAL loads from a clamped address (validity is applied when the values are consumed): a load inside a branch is
            // merged with the "else 0" by moves right behind it, i.e. waited for at once — six serial round trips instead of six in flight
            const int Y = min(max(iy0 + py, 0), H - 1), X = min(max(ix0 + px, 0), W - 1);
            const SRC* sp = src + ((size_t)Y * W + X) * C;
            pre[r][0] = (RAW)sp[0], pre[r][1] = (RAW)sp[1], pre[r][2] = (RAW)sp[2];
        }
    };
    int item = blockIdx.x;
    if (item >= n_items) return;
    prefetch(item);
    for (; item < n_items; item += gridDim.x) {
        int f, Y0, X0;
        decode(item, f, Y0, X0);
        float* const P = fb.P[f];
        const int iy0 = Y0 - 3, ix0 = X0 - 3;
        // ---- phase 0: registers -> LDS (clamped / scaled), plane 0 of the tile -> HBM
#pragma unroll
        for (int r = 0; r < ENC_NPRE; ++r) {
            const int i = tid + r * 256;
            // the prefetched registers become visible to the optimiser HERE: without the pin hipcc computes the clamp / division right
            // behind the loads of the previous iteration (pure arithmetic on loaded values) and waits for them there — no prefetch
            RAW x0 = pre[r][0], x1 = pre[r][1], x2 = pre[r][2];
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
            if (i < ENC_TI * ENC_TI) {
                const int py = i / ENC_TI, px = i - py * ENC_TI;
                const int Y = iy0 + py, X = ix0 + px;
                float4 v = {0.f, 0.f, 0.f, 0.f};
                if (Y >= 0 && Y < H && X >= 0 && X < W) {
                    if (sizeof(SRC) == 1) {
                        v.x = __fdiv_rn((float)x0, 255.0f);
                        v.y = __fdiv_rn((float)x1, 255.0f);
                        v.z = __fdiv_rn((float)x2, 255.0f);
                    } else {
                        v.x = fminf(fmaxf((float)x0, 0.f), 1.f);
                        v.y = fminf(fmaxf((float)x1, 0.f), 1.f);
                        v.z = fminf(fmaxf((float)x2, 0.f), 1.f);
                    }
                }
                sI[i] = v;
                if (py >= 3 && py < 3 + ENC_T && px >= 3 && px < 3 + ENC_T && Y < Hp && X < Wp) *(float4*)(P + ((size_t)Y * Wp + X) * 4) = v;
            }
        }
        // the next item's source tile: in flight under phases 1 and 2.  Unconditional (the last item re-reads its own tile): behind a
        // branch the loaded registers are merged with the old ones by moves at the end of the block — and waited for there
        prefetch(min(item + (int)gridDim.x, n_items - 1));
        enc_lds_barrier();
        // ---- phase 1: E tile
        const int ey0 = Y0 / 2 - 1, ex0 = X0 / 2 - 1;
        for (int i = tid; i < ENC_TE * ENC_TE; i += 256) {
            const int py = i / ENC_TE, px = i - py * ENC_TE;
            const int ey = ey0 + py, ex = ex0 + px;
            float acc[CM];
#pragma unroll
            for (int co = 0; co < CM; ++co) acc[co] = 0.f;
            const bool in_img = ey >= 0 && ey < He && ex >= 0 && ex < We;
            if (in_img) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 v = sI[(2 * py + ky) * ENC_TI + 2 * px + kx];
                        const float* wt = w0 + (ky * 3 + kx) * 3 * CM;
#pragma unroll
                        for (int co = 0; co < CM; ++co) acc[co] = fmaf(v.z, wt[2 * CM + co], fmaf(v.y, wt[CM + co], fmaf(v.x, wt[co], acc[co])));
                    }
#pragma unroll
                for (int co = 0; co < CM; ++co) acc[co] += b0[co];
            }
#pragma unroll
            for (int q = 0; q < CM / 4; ++q) *(float4*)&sE[i * CM + 4 * q] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
        enc_lds_barrier();
        // ---- phase 2: one E pixel of the 16x16 centre per thread -> its 2x2 output quad
        {
            const int lx = tid & 15, ly = tid >> 4;
            const int y = Y0 / 2 + ly, x = X0 / 2 + lx;
            if (y < He && x < We) {
                float acc[4][CF];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int co = 0; co < CF; ++co) acc[g][co] = b1[co];
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const float* e = &sE[((ly + 1 + dy) * ENC_TE + lx + 1 + dx) * CM];
#pragma unroll
                        for (int q = 0; q < CM / 4; ++q) {
                            const float4 t = *(const float4*)(e + 4 * q);
                            const float ev[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                            for (int py = 0; py < 2; ++py) {
                                if (dy < py - 1 || dy > py) continue;
                                const int ky = py + 1 - 2 * dy;
#pragma unroll
                                for (int px = 0; px < 2; ++px) {
                                    if (dx < px - 1 || dx > px) continue;
                                    const int kx = px + 1 - 2 * dx;
                                    const float* wt = w1 + ((ky * 4 + kx) * CM + 4 * q) * CF;
#pragma unroll
                                    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                                        for (int co = 0; co < CF; ++co) acc[py * 2 + px][co] = fmaf(ev[ci], wt[ci * CF + co], acc[py * 2 + px][co]);
                                }
                            }
                        }
                    }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int Y = 2 * y + (g >> 1), X = 2 * x + (g & 1);
                    *(float4*)(P + (size_t)Hp * Wp * 4 + ((size_t)Y * Wp + X) * 4) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
                }
            }
        }
        enc_lds_barrier();      // sE (and sI) are rewritten by the next item
    }
}

// n frames (<= ENC_MAXF per launch; longer lists go out in several launches): srcs[i] -> packs[i]
int encode47_batch_launch(int n, const void* const* srcs, bool u8, float* const* packs, const float* w0, const float* b0, const float* w1,
                          const float* b1, int H, int W, int C, int Hp, int Wp, hipStream_t s) {
    const int tiles_x = cdiv(Wp, ENC_T), tiles_y = cdiv(Hp, ENC_T);
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t pr;
    static int cus_cache[kMaxDevices] = {};
    if (dev >= 0 && dev < kMaxDevices && !cus_cache[dev]) {
        VFI_CHECK_HIP(hipGetDeviceProperties(&pr, dev));
        cus_cache[dev] = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    const int cus = (dev >= 0 && dev < kMaxDevices) ? cus_cache[dev] : 256;
    for (int base = 0; base < n; base += ENC_MAXF) {
        const int m = n - base < ENC_MAXF ? n - base : ENC_MAXF;
        EncodeBatch fb = {};
        for (int i = 0; i < m; ++i) fb.src[i] = srcs[base + i], fb.P[i] = packs[base + i];
        const int per = tiles_x * tiles_y, items = per * m;
        const int grid = items < 3 * launch_cus(cus) ? items : 3 * launch_cus(cus);      // 42.6 KB of LDS: three workgroups per CU
        TraceScope ts("encode_batch", s);
        if (u8)
            hipLaunchKernelGGL(encode47_batch_kernel<unsigned char>, dim3(grid), dim3(256), 0, s, fb, w0, b0, w1, b1, H, W, C, Hp, Wp, tiles_x, per, items);
        else
            hipLaunchKernelGGL(encode47_batch_kernel<float>, dim3(grid), dim3(256), 0, s, fb, w0, b0, w1, b1, H, W, C, Hp, Wp, tiles_x, per, items);
        VFI_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

int prep_frame_launch(const float* src, float* P, int H, int W, int C, int Hp, int Wp, hipStream_t s) {
    TraceScope ts("prep_frame", s);
    hipLaunchKernelGGL(prep_frame_kernel<float>, dim3(cdiv(Hp * Wp, 256)), dim3(256), 0, s, src, P, H, W, C, Hp, Wp);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int prep_frame_u8_launch(const unsigned char* src, float* P, int H, int W, int C, int Hp, int Wp, hipStream_t s) {
    TraceScope ts("prep_frame", s);
    hipLaunchKernelGGL(prep_frame_kernel<unsigned char>, dim3(cdiv(Hp * Wp, 256)), dim3(256), 0, s, src, P, H, W, C, Hp, Wp);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int f32_to_u8_launch(const float* in, unsigned char* out, long n, hipStream_t s) {
    TraceScope ts("f32_to_u8", s);
    hipLaunchKernelGGL(f32_to_u8_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, in, out, n);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int encode_conv_launch(const float* P, float* E, const float* w0, const float* b0, int CM, bool act, int Hp, int Wp,
                       hipStream_t s) {
    TraceScope ts("encode_conv", s);
    const dim3 grid(cdiv(Hp / 2 * (Wp / 2), 128));
    if (CM == 16 && !act)
        hipLaunchKernelGGL((encode_conv_kernel<16, false>), grid, dim3(128), 0, s, P, w0, b0, E, Hp, Wp);
    else if (CM == 16 && act)
        hipLaunchKernelGGL((encode_conv_kernel<16, true>), grid, dim3(128), 0, s, P, w0, b0, E, Hp, Wp);
    else if (CM == 32 && act)
        hipLaunchKernelGGL((encode_conv_kernel<32, true>), grid, dim3(128), 0, s, P, w0, b0, E, Hp, Wp);
    else
        VFI_REQUIRE(false, "encode_conv: unsupported head (CM=%d act=%d)", CM, (int)act);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int encode_deconv_launch(const float* E, float* P, const float* w1, const float* b1, int CM, int CF, int Hp, int Wp,
                         hipStream_t s) {
    TraceScope ts("encode_deconv", s);
    const dim3 grid(cdiv(Hp / 2 * (Wp / 2), 128));
    if (CM == 16 && CF == 4)
        hipLaunchKernelGGL((encode_deconv_kernel<16, 4>), grid, dim3(128), 0, s, E, w1, b1, P, Hp, Wp);
    else if (CM == 32 && CF == 8)
        hipLaunchKernelGGL((encode_deconv_kernel<32, 8>), grid, dim3(128), 0, s, E, w1, b1, P, Hp, Wp);
    else
        VFI_REQUIRE(false, "encode_deconv: unsupported head (CM=%d CF=%d)", CM, CF);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// stage input: torch.cat(...) + F.interpolate(x, 1/s) (+ interpolate(flow)/s) fused.
// Down-resizing by an even integer s with align_corners=False samples exactly the centre 2x2 of
// every s x s cell with weights 1/2, so only those pixels are warped at all.
// ---------------------------------------------------------------------------------------
template <bool HAS_FLOW, int NP, int NF, int NX>
__global__ __launch_bounds__(128) void stage_in_kernel(const float* __restrict__ Ppool, size_t pack_stride,
                                                       RifeTasks tasks, const float* __restrict__ F,
                                                       const float* __restrict__ M, const float* __restrict__ FEAT,
                                                       float* __restrict__ Xo, int Hp, int Wp, int s, int CX) {
    const int Hs = Hp / s, Ws = Wp / s;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hs * Ws) return;
    const int b = blockIdx.y;
    const int xl = idx % Ws, yl = idx / Ws;
    const float* P0 = Ppool + (size_t)tasks.slot0[b] * pack_stride;
    const float* P1 = Ppool + (size_t)tasks.slot1[b] * pack_stride;
    const size_t hi_off = (size_t)Hp * Wp * 4;
    const float tstep = tasks.t[b];
    const WarpGeo g = make_warp_geo(Wp, Hp);
    const int off = NP == 1 ? 0 : s / 2 - 1;
    const float inv_s = 1.0f / (float)s;
    constexpr int NI = 7 + 8 * NF;                    // images + features + timestep
    constexpr int NC = HAS_FLOW ? NI + 5 + NX : NI;   // + mask (+ NX carried block features, arch 4.26) + flow
    static_assert(NX == 0 || (NX == 8 && HAS_FLOW), "carried features come with a flow");
    constexpr int NR = (NC + 7) / 8 * 8;         // X channels (zero padded)

    // torch upsample_bilinear2d order: wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d), all weights 0.5
    float r[NR], row[NC], o[NC];
#pragma unroll
    for (int c = 0; c < NR; ++c) r[c] = 0.f;
#pragma unroll 1
    for (int k = 0; k < NP * NP; ++k) {  // rolled on purpose: keeps the gather's register footprint small
        const int dy = k / NP, dx = k % NP;
        {
            const int Y = yl * s + off + dy, X = xl * s + off + dx;
            const size_t p = (size_t)Y * Wp + X;
            float4 a_lo, b_lo, a_hi[NF], b_hi[NF];
            if (HAS_FLOW) {
                const size_t pb = (size_t)b * Hp * Wp + p;
                const float4 f = ((const float4*)F)[pb];
                const Tap4 t0 = warp_taps(g, X, Y, f.x, f.y);
                const Tap4 t1 = warp_taps(g, X, Y, f.z, f.w);
                sample_pack<NF>(P0, hi_off, t0, a_lo, a_hi);
                sample_pack<NF>(P1, hi_off, t1, b_lo, b_hi);
                o[NI] = M[pb];
                if (NX == 8) {
                    const float4 g0 = ((const float4*)FEAT)[(size_t)b * 2 * Hp * Wp + p];
                    const float4 g1 = ((const float4*)FEAT)[((size_t)b * 2 + 1) * Hp * Wp + p];
                    o[NI + 1] = g0.x; o[NI + 2] = g0.y; o[NI + 3] = g0.z; o[NI + 4] = g0.w;
                    o[NI + 5] = g1.x; o[NI + 6] = g1.y; o[NI + 7] = g1.z; o[NI + 8] = g1.w;
                }
                o[NC - 4] = f.x;
                o[NC - 3] = f.y;
                o[NC - 2] = f.z;
                o[NC - 1] = f.w;
            } else {
                a_lo = ((const float4*)P0)[p];
                b_lo = ((const float4*)P1)[p];
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    a_hi[j] = ((const float4*)(P0 + (size_t)(1 + j) * hi_off))[p];
                    b_hi[j] = ((const float4*)(P1 + (size_t)(1 + j) * hi_off))[p];
                }
            }
            cat_inputs<NF>(o, a_lo, b_lo, a_hi, b_hi, tstep);
#pragma unroll
            for (int c = 0; c < NC; ++c) row[c] = dx == 0 ? o[c] : __fadd_rn(0.5f * row[c], 0.5f * o[c]);
        }
        if (dx == NP - 1) {
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = dy == 0 ? row[c] : __fadd_rn(0.5f * r[c], 0.5f * row[c]);
        }
    }
    if (HAS_FLOW) {
#pragma unroll
        for (int c = NC - 4; c < NC; ++c) r[c] = r[c] * inv_s;  // interpolate(flow) * 1.0 / scale
    }
    float4* op = (float4*)Xo + (size_t)b * (CX / 4) * Hs * Ws + idx;
#pragma unroll
    for (int q = 0; q < NR / 4; ++q)
        if (q < CX / 4) op[(size_t)q * Hs * Ws] = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
}

int stage_in_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* F,
                    const float* M, const float* FEAT, float* X, int Hp, int Wp, int s, int CX, int NF, bool has_flow,
                    hipStream_t st) {
    const int Hs = Hp / s, Ws = Wp / s;
    const int NX = FEAT && has_flow ? 8 : 0;
    VFI_REQUIRE(s == 1 || s % 2 == 0, "stage_in: scale %d must be 1 or even", s);
    VFI_REQUIRE((NF == 1 || NF == 2) && (NX == 0 || NF == 1) && CX == round_up(7 + 8 * NF + (has_flow ? 5 + NX : 0), 8),
                "stage_in: bad CX %d for %d feature planes (+%d carried)", CX, NF, NX);
    dim3 grid(cdiv(Hs * Ws, 128), B);
    TraceScope ts(has_flow ? "stage_in_warp" : "stage_in0", st);
#define VFI_SI(HF, NPV, NFV, NXV) \
    hipLaunchKernelGGL((stage_in_kernel<HF, NPV, NFV, NXV>), grid, dim3(128), 0, st, Ppool, pack_stride, tasks, F, M, FEAT, X, Hp, Wp, s, CX)
#define VFI_SI2(HF, NPV) \
    do { if (NF == 1) VFI_SI(HF, NPV, 1, 0); else VFI_SI(HF, NPV, 2, 0); } while (0)
    if (NX == 8) {
        if (s == 1) VFI_SI(true, 1, 1, 8); else VFI_SI(true, 2, 1, 8);
    } else if (has_flow) {
        if (s == 1) VFI_SI2(true, 1); else VFI_SI2(true, 2);
    } else {
        if (s == 1) VFI_SI2(false, 1); else VFI_SI2(false, 2);
    }
#undef VFI_SI2
#undef VFI_SI
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// block output -> full resolution:  tmp = interpolate(PixelShuffle(deconv), x s);
// flow (+)= tmp[:, :4]*s; mask = tmp[:, 4:5]                       rife_arch.py:262-276,645,698
// ---------------------------------------------------------------------------------------
// T is planar4 [2][Hs][Ws][4]: flow delta = plane 0, mask = plane 1 component 0
struct TVal {
    float4 f;
    float m;
};
__device__ static inline TVal t_read(const float* __restrict__ Tb, int Hs, int Ws, int Yt, int Xt) {
    const size_t p = (size_t)Yt * Ws + Xt;
    TVal v;
    v.f = ((const float4*)Tb)[p];
    v.m = Tb[((size_t)Hs * Ws + p) * 4];
    return v;
}
__device__ static inline TVal t_bilerp(const TVal& a, const TVal& b, const TVal& c, const TVal& d, float wy0, float wy1,
                                       float wx0, float wx1) {
    // torch upsample_bilinear2d: wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d), explicit rounding order
#define VFI_BL(A, B, C, D) \
    __fadd_rn(__fmul_rn(wy0, __fadd_rn(__fmul_rn(wx0, A), __fmul_rn(wx1, B))), __fmul_rn(wy1, __fadd_rn(__fmul_rn(wx0, C), __fmul_rn(wx1, D))))
    TVal r;
    r.f.x = VFI_BL(a.f.x, b.f.x, c.f.x, d.f.x);
    r.f.y = VFI_BL(a.f.y, b.f.y, c.f.y, d.f.y);
    r.f.z = VFI_BL(a.f.z, b.f.z, c.f.z, d.f.z);
    r.f.w = VFI_BL(a.f.w, b.f.w, c.f.w, d.f.w);
    r.m = VFI_BL(a.m, b.m, c.m, d.m);
#undef VFI_BL
    return r;
}

struct Bil {
    int i0, i1;
    float w0, w1;
};
// torch area_pixel_compute_source_index(align_corners=False) + guard_index_and_lambda
__device__ static inline Bil bil_index(int d, float rscale, int in_size) {
    float src = __fsub_rn(__fmul_rn(rscale, __fadd_rn((float)d, 0.5f)), 0.5f);
    if (src < 0.f) src = 0.f;
    int i0 = (int)floorf(src);
    if (i0 > in_size - 1) i0 = in_size - 1;
    float l = __fsub_rn(src, (float)i0);
    l = fminf(fmaxf(l, 0.f), 1.f);
    Bil b;
    b.i0 = i0;
    b.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    b.w1 = l;
    b.w0 = __fsub_rn(1.0f, l);
    return b;
}

__device__ static inline TVal t_upsample(const float* __restrict__ Tb, int Hs, int Ws, int s, int Y, int X) {
    if (s == 1) return t_read(Tb, Hs, Ws, Y, X);
    const float rs = 1.0f / (float)s;
    const Bil by = bil_index(Y, rs, Hs), bx = bil_index(X, rs, Ws);
    return t_bilerp(t_read(Tb, Hs, Ws, by.i0, bx.i0), t_read(Tb, Hs, Ws, by.i0, bx.i1), t_read(Tb, Hs, Ws, by.i1, bx.i0),
                    t_read(Tb, Hs, Ws, by.i1, bx.i1), by.w0, by.w1, bx.w0, bx.w1);
}

template <bool HAS_PREV>
__global__ void flow_up_kernel(const float* __restrict__ T, float* __restrict__ F, float* __restrict__ M, int Hp,
                               int Wp, int s, int tp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hp * Wp) return;
    const int b = blockIdx.y;
    const int X = idx % Wp, Y = idx / Wp;
    const int Hs = Hp / s, Ws = Wp / s;
    const float* Tb = T + (size_t)b * Hs * Ws * 4 * tp;
    const TVal v = t_upsample(Tb, Hs, Ws, s, Y, X);
    const size_t pb = (size_t)b * Hp * Wp + idx;
    const float fs = (float)s;
    float4 f = make_float4(v.f.x * fs, v.f.y * fs, v.f.z * fs, v.f.w * fs);
    if (HAS_PREV) {
        const float4 o = ((const float4*)F)[pb];
        f = make_float4(o.x + f.x, o.y + f.y, o.z + f.z, o.w + f.w);
    }
    ((float4*)F)[pb] = f;
    M[pb] = v.m;
}

int flow_up_launch(const float* T, float* F, float* M, int B, int Hp, int Wp, int s, int tp, bool has_prev,
                   hipStream_t st) {
    dim3 grid(cdiv(Hp * Wp, 256), B);
    TraceScope ts("flow_up", st);
    if (has_prev)
        hipLaunchKernelGGL(flow_up_kernel<true>, grid, dim3(256), 0, st, T, F, M, Hp, Wp, s, tp);
    else
        hipLaunchKernelGGL(flow_up_kernel<false>, grid, dim3(256), 0, st, T, F, M, Hp, Wp, s, tp);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// fused block transition i -> i+1 for the standard scale list (block scales 2*SP -> SP):
//   flow_up(i)  +  stage_in(i+1)  in one pass over the full-resolution flow field.
// One thread owns one SP x SP cell (= one pixel of X_{i+1}).  Because SP divides the previous scale
// 2*SP, every pixel of the cell up-samples from the same 2x2 T pixels, so the 20 T values are loaded
// once per thread; the mask never goes to memory and the flow is read+written exactly once.
// Thread blocks are 16x16 cells (2-D) so that warp taps of vertically adjacent pixels share L2 lines.
// ---------------------------------------------------------------------------------------
template <int SP, bool HAS_PREV, int NF>
__global__ __launch_bounds__(256) void stage_trans_kernel(const float* __restrict__ Ppool, size_t pack_stride,
                                                          RifeTasks tasks, const float* __restrict__ T,
                                                          float* __restrict__ F, float* __restrict__ Xo, int Hp,
                                                          int Wp, int tiles_x) {
    constexpr int SI = 2 * SP;            // scale of the block that produced T
    constexpr int NP = SP == 1 ? 1 : 2;   // centre pixels per axis that the down-resize samples
    constexpr int OFF = SP == 1 ? 0 : SP / 2 - 1;
    const int Hs = Hp / SP, Ws = Wp / SP;
    const int b = blockIdx.y;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int xl = tile_x * 16 + (threadIdx.x & 15), yl = tile_y * 16 + (threadIdx.x >> 4);
    if (xl >= Ws || yl >= Hs) return;
    const int Hi = Hp / SI, Wi = Wp / SI;  // resolution of the pixel-shuffled T
    const float* Tb = T + (size_t)b * Hi * Wi * 8;
    const float rs = 1.0f / (float)SI;
    const Bil by0 = bil_index(yl * SP, rs, Hi), bx0 = bil_index(xl * SP, rs, Wi);
    const TVal t00 = t_read(Tb, Hi, Wi, by0.i0, bx0.i0), t01 = t_read(Tb, Hi, Wi, by0.i0, bx0.i1);
    const TVal t10 = t_read(Tb, Hi, Wi, by0.i1, bx0.i0), t11 = t_read(Tb, Hi, Wi, by0.i1, bx0.i1);
    // ---- phase 1: flow (+)= up(T)*SI for every pixel of the cell; keep flow+mask of the centre pixels
    float4 fc[NP * NP];
    float mc[NP * NP];
#pragma unroll
    for (int dy = 0; dy < SP; ++dy) {
        const Bil wy = bil_index(yl * SP + dy, rs, Hi);
#pragma unroll
        for (int dx = 0; dx < SP; ++dx) {
            const Bil wx = bil_index(xl * SP + dx, rs, Wi);
            const TVal v = t_bilerp(t00, t01, t10, t11, wy.w0, wy.w1, wx.w0, wx.w1);
            const size_t pb = (size_t)b * Hp * Wp + (size_t)(yl * SP + dy) * Wp + xl * SP + dx;
            const float fs = (float)SI;
            float4 f = make_float4(v.f.x * fs, v.f.y * fs, v.f.z * fs, v.f.w * fs);
            if (HAS_PREV) {
                const float4 o = ((const float4*)F)[pb];
                f = make_float4(o.x + f.x, o.y + f.y, o.z + f.z, o.w + f.w);
            }
            ((float4*)F)[pb] = f;
            const int cy = dy - OFF, cx = dx - OFF;
            if (cy >= 0 && cy < NP && cx >= 0 && cx < NP) {
                fc[cy * NP + cx] = f;
                mc[cy * NP + cx] = v.m;
            }
        }
    }
    // ---- phase 2: warp both frame packs at the centre pixels, down-resize (weights 1/2), write X
    const float* P0 = Ppool + (size_t)tasks.slot0[b] * pack_stride;
    const float* P1 = Ppool + (size_t)tasks.slot1[b] * pack_stride;
    const size_t hi_off = (size_t)Hp * Wp * 4;
    const float tstep = tasks.t[b];
    const WarpGeo g = make_warp_geo(Wp, Hp);
    constexpr int NC = 12 + 8 * NF;          // images, features, timestep, mask, flow
    constexpr int NR = (NC + 7) / 8 * 8;     // X channels (zero padded)
    float r[NR], row[NC], o[NC];
#pragma unroll
    for (int c = 0; c < NR; ++c) r[c] = 0.f;
#pragma unroll 1
    for (int k = 0; k < NP * NP; ++k) {  // rolled on purpose (register footprint of the gathers)
        const int dy = k / NP, dx = k % NP;
        float4 f = fc[0];
        float m = mc[0];
#pragma unroll
        for (int q = 1; q < NP * NP; ++q) {  // select without dynamic register indexing
            if (k == q) {
                f = fc[q];
                m = mc[q];
            }
        }
        const int Y = yl * SP + OFF + dy, X = xl * SP + OFF + dx;
        const Tap4 t0 = warp_taps(g, X, Y, f.x, f.y);
        const Tap4 t1 = warp_taps(g, X, Y, f.z, f.w);
        float4 a_lo, b_lo, a_hi[NF], b_hi[NF];
        sample_pack<NF>(P0, hi_off, t0, a_lo, a_hi);
        sample_pack<NF>(P1, hi_off, t1, b_lo, b_hi);
        cat_inputs<NF>(o, a_lo, b_lo, a_hi, b_hi, tstep);
        o[NC - 5] = m;
        o[NC - 4] = f.x; o[NC - 3] = f.y; o[NC - 2] = f.z; o[NC - 1] = f.w;
#pragma unroll
        for (int c = 0; c < NC; ++c) row[c] = dx == 0 ? o[c] : __fadd_rn(0.5f * row[c], 0.5f * o[c]);
        if (dx == NP - 1) {
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = dy == 0 ? row[c] : __fadd_rn(0.5f * r[c], 0.5f * row[c]);
        }
    }
    const float inv_s = 1.0f / (float)SP;
#pragma unroll
    for (int c = NC - 4; c < NC; ++c) r[c] = r[c] * inv_s;  // interpolate(flow) * 1.0 / scale
    float4* op = (float4*)Xo + (size_t)b * (NR / 4) * Hs * Ws + (size_t)yl * Ws + xl;
#pragma unroll
    for (int q = 0; q < NR / 4; ++q) op[(size_t)q * Hs * Ws] = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
}

// ---------------------------------------------------------------------------------------
// The same transition with one thread per warped pixel (SP = 2, 4, 8).  The cell-per-thread kernel above touches F with
// a lane stride of SP pixels (SP store instructions per row, each filling 1/SP of the lines it touches) and walks its
// 4 centre pixels one after the other; at SP = 2 every pixel is a centre pixel and that kernel ran at ~2.2 TB/s.
// Here the 4 lanes of a quad are the 2x2 centre pixels of one cell:
//   * F is updated with row-contiguous accesses (SP >= 4: a separate pass over the (16 SP)x(4 SP)-pixel tile, the centre
//     pixels' flow/mask/features handed to their quad lanes through 5-13 KB of LDS; SP = 2: the quad lanes themselves);
//   * the 2x2 T pixels that a cell up-samples from are loaded one corner per lane and exchanged with quad DPP moves;
//   * the down-resize (weights 1/2, torch's row-then-column order) is two quad swaps per channel;
//   * each lane of the quad stores a different planar4 plane of X.
// Arithmetic is expression-for-expression the cell kernel's; results agree to rounding noise (hipcc chooses the FMA
// contractions of the bilinear expressions per kernel: max 1.5e-5 on the output, tests/test_gpu_rife.py).
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ static inline float quad_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// One pixel of the block output T as 5+NX floats: flow delta 4, mask, then (arch 4.26, NX = 8) the carried features.
// NX = 0: planar4 [2][Hs][Ws][4] (plane 1 = mask,-,-,-); NX = 8: [4][Hs][Ws][4] (mask,g0..g2 | g3..g6 | g7,-,-,-).
// FULL = false reads the flow plane only.
template <int NX, bool FULL>
__device__ static inline void tq_read(const float* __restrict__ Tb, int Hs, int Ws, int Yt, int Xt, float* v) {
    const size_t p = (size_t)Yt * Ws + Xt, ps = (size_t)Hs * Ws;
    const float4 f = ((const float4*)Tb)[p];
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    if constexpr (FULL) {
        if constexpr (NX == 0) {
            v[4] = Tb[(ps + p) * 4];
        } else {
            const float4 p1 = ((const float4*)Tb)[ps + p], p2 = ((const float4*)Tb)[2 * ps + p];
            v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
            v[8] = p2.x; v[9] = p2.y; v[10] = p2.z; v[11] = p2.w;
            v[4 + NX] = Tb[(3 * ps + p) * 4];
        }
    }
}
// up-sampled T at pixel (X,Y); the quad's 4 lanes must share the 2x2 source pixels (same cell row/column run):
// lane ql fetches corner (ql>>1, ql&1), quad broadcasts (quad_perm [j,j,j,j]) deliver the other three.
template <int NX, bool FULL>
__device__ static inline void t_upsample_quad(const float* __restrict__ Tb, int Hi, int Wi, float rs, int Y, int X, int ql,
                                              float* out) {
    constexpr int N = FULL ? 5 + NX : 4;
    const Bil by = bil_index(Y, rs, Hi), bx = bil_index(X, rs, Wi);
    float mine[5 + NX];
    tq_read<NX, FULL>(Tb, Hi, Wi, (ql >> 1) ? by.i1 : by.i0, (ql & 1) ? bx.i1 : bx.i0, mine);
    const float wy0 = by.w0, wy1 = by.w1, wx0 = bx.w0, wx1 = bx.w1;
    // torch upsample_bilinear2d: wy0*(wx0*a + wx1*b) + wy1*(wx0*c + wx1*d), as t_bilerp
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const float a = quad_f<0x00>(mine[j]), bb = quad_f<0x55>(mine[j]), c = quad_f<0xAA>(mine[j]), d = quad_f<0xFF>(mine[j]);
        out[j] = __fadd_rn(__fmul_rn(wy0, __fadd_rn(__fmul_rn(wx0, a), __fmul_rn(wx1, bb))),
                           __fmul_rn(wy1, __fadd_rn(__fmul_rn(wx0, c), __fmul_rn(wx1, d))));
    }
}

template <int SP, bool HAS_PREV, int NF, int NX>
__global__ __launch_bounds__(256) void stage_trans_quad_kernel(const float* __restrict__ Ppool, size_t pack_stride,
                                                               RifeTasks tasks, const float* __restrict__ T,
                                                               float* __restrict__ F, float* __restrict__ Xo, int Hp,
                                                               int Wp, int tiles_x, int xcd_per, int n_tiles) {
    static_assert(SP == 2 || SP == 4 || SP == 8, "quad transition: block scales 4->2, 8->4, 16->8");
    // xcd_per > 0: XCD-aware order (workgroup x runs on XCD x % 8: XCD k takes the contiguous band of tiles [k * xcd_per, (k + 1) * xcd_per),
    // so that the pack rows neighbouring tile rows share are read through one L2)
    const int bid = xcd_per > 0 ? ((int)blockIdx.x & 7) * xcd_per + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    if (bid >= n_tiles || (xcd_per > 0 && ((int)blockIdx.x >> 3) >= xcd_per)) return;
    static_assert(NX == 0 || (NX == 8 && NF == 1), "carried block features: arch 4.26 only");
    constexpr int SI = 2 * SP;          // scale of the block that produced T
    constexpr int OFF = SP / 2 - 1;     // first centre pixel of a cell
    constexpr int NV = 5 + NX;          // flow 4, mask, carried features
    const int Hs = Hp / SP, Ws = Wp / SP, Hi = Hp / SI, Wi = Wp / SI;
    const int b = blockIdx.y;
    const int tile_y = bid / tiles_x, tile_x = bid - tile_y * tiles_x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ql = lane & 3;
    const float* Tb = T + (size_t)b * Hi * Wi * (NX ? 16 : 8);
    const float rs = 1.0f / (float)SI, fs = (float)SI;
    // block tile = 16 x 4 cells; a wave is one row of 16 cells, a quad one cell
    const int xl = tile_x * 16 + (lane >> 2), yl = tile_y * 4 + wave;
    const int X = xl * SP + OFF + (ql & 1), Y = yl * SP + OFF + (ql >> 1);
    const bool cell_ok = xl < Ws && yl < Hs;
    float v[NV];   // this centre pixel: updated flow, mask, carried features
    if constexpr (SP == 2) {
        if (!cell_ok) return;   // whole quads leave together
        t_upsample_quad<NX, true>(Tb, Hi, Wi, rs, Y, X, ql, v);
        const size_t pb = (size_t)b * Hp * Wp + (size_t)Y * Wp + X;
        float4 f = make_float4(v[0] * fs, v[1] * fs, v[2] * fs, v[3] * fs);
        if (HAS_PREV) {
            const float4 o = ((const float4*)F)[pb];
            f = make_float4(o.x + f.x, o.y + f.y, o.z + f.z, o.w + f.w);
        }
        ((float4*)F)[pb] = f;
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
        // ---- phase 1: flow (+)= up(T)*SI over the (16*SP)x(4*SP)-pixel tile, 64 contiguous pixels per wave and pass;
        // a quad is 4 pixels of one cell row, so it shares its 2x2 T pixels
        __shared__ float centre[NV][256];
#pragma unroll
        for (int p = 0; p < SP; ++p) {
            constexpr int NH = SP / 4;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int col = h * 64 + lane;
                const int Yp = yl * SP + p, Xp = tile_x * 16 * SP + col;
                if (Xp < Wp && Yp < Hp) {
                    const bool crow = p == OFF || p == OFF + 1;   // a row with centre pixels: mask / features too
                    float u[NV];
                    if (crow) t_upsample_quad<NX, true>(Tb, Hi, Wi, rs, Yp, Xp, ql, u);
                    else t_upsample_quad<NX, false>(Tb, Hi, Wi, rs, Yp, Xp, ql, u);
                    const size_t pb = (size_t)b * Hp * Wp + (size_t)Yp * Wp + Xp;
                    float4 fp = make_float4(u[0] * fs, u[1] * fs, u[2] * fs, u[3] * fs);
                    if (HAS_PREV) {
                        const float4 o = ((const float4*)F)[pb];
                        fp = make_float4(o.x + fp.x, o.y + fp.y, o.z + fp.z, o.w + fp.w);
                    }
                    ((float4*)F)[pb] = fp;
                    const int ix = col & (SP - 1);
                    if (crow && (ix == OFF || ix == OFF + 1)) {   // a centre pixel: hand over to its quad lane
                        const int dst = wave * 64 + (col / SP) * 4 + (ix - OFF) + ((p - OFF) << 1);
                        centre[0][dst] = fp.x; centre[1][dst] = fp.y; centre[2][dst] = fp.z; centre[3][dst] = fp.w;
#pragma unroll
                        for (int j = 4; j < NV; ++j) centre[j][dst] = u[j];
                    }
                }
            }
        }
        __syncthreads();
        if (!cell_ok) return;
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = centre[j][threadIdx.x];
    }
    // ---- phase 2: warp both frame packs at this centre pixel, down-resize across the quad, write X
    const float* P0 = Ppool + (size_t)tasks.slot0[b] * pack_stride;
    const float* P1 = Ppool + (size_t)tasks.slot1[b] * pack_stride;
    const size_t hi_off = (size_t)Hp * Wp * 4;
    const WarpGeo g = make_warp_geo(Wp, Hp);
    constexpr int NC = 12 + 8 * NF + NX;     // images, features, timestep, mask, carried features, flow
    constexpr int NR = (NC + 7) / 8 * 8;     // X channels (zero padded)
    float r[NR];
    {
        const Tap4 t0 = warp_taps(g, X, Y, v[0], v[1]);
        const Tap4 t1 = warp_taps(g, X, Y, v[2], v[3]);
        float4 a_lo, b_lo, a_hi[NF], b_hi[NF];
        sample_pack<NF>(P0, hi_off, t0, a_lo, a_hi);
        sample_pack<NF>(P1, hi_off, t1, b_lo, b_hi);
        cat_inputs<NF>(r, a_lo, b_lo, a_hi, b_hi, tasks.t[b]);
#pragma unroll
        for (int j = 4; j < NV; ++j) r[3 + 8 * NF + j] = v[j];   // mask at 7+8NF, carried features after it
        r[NC - 4] = v[0]; r[NC - 3] = v[1]; r[NC - 2] = v[2]; r[NC - 1] = v[3];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {   // 0.5*left + 0.5*right, then 0.5*top + 0.5*bottom (all four lanes get the result)
        const float row = __fadd_rn(0.5f * r[c], 0.5f * quad_f<0xB1>(r[c]));
        r[c] = __fadd_rn(0.5f * row, 0.5f * quad_f<0x4E>(row));
    }
#pragma unroll
    for (int c = NC; c < NR; ++c) r[c] = 0.f;
    const float inv_s = 1.0f / (float)SP;
#pragma unroll
    for (int c = NC - 4; c < NC; ++c) r[c] = r[c] * inv_s;  // interpolate(flow) * 1.0 / scale
    float4* op = (float4*)Xo + (size_t)b * (NR / 4) * Hs * Ws + (size_t)yl * Ws + xl;
#pragma unroll
    for (int base = 0; base < NR / 4; base += 4) {   // lane ql of the quad stores plane base+ql
        float4 val = make_float4(r[4 * base], r[4 * base + 1], r[4 * base + 2], r[4 * base + 3]);
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (base + j < NR / 4 && ql == j)
                val = make_float4(r[4 * (base + j)], r[4 * (base + j) + 1], r[4 * (base + j) + 2], r[4 * (base + j) + 3]);
        if (base + ql < NR / 4) op[(size_t)(base + ql) * Hs * Ws] = val;
    }
}

// option stage_quad: bit mask of the next-block scales (2, 4, 8) that take the quad kernel; default all
static int stage_quad_mask() { return (int)option(kOptStageQuad); }

int stage_trans_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* T, float* F,
                       float* X, int Hp, int Wp, int s_prev, int s_next, int NF, bool has_prev, hipStream_t st) {
    VFI_REQUIRE(s_prev == 2 * s_next && (s_next == 4 || s_next == 2 || s_next == 1) && (NF == 1 || NF == 2),
                "stage_trans: scales %d -> %d (%d feature planes) not on the fused path", s_prev, s_next, NF);
    const int Hs = Hp / s_next, Ws = Wp / s_next;
    const int tiles_x = cdiv(Ws, 16), tiles_y = cdiv(Hs, 16);
    dim3 grid(tiles_x * tiles_y, B);
    TraceScope ts(s_next == 4 ? "stage_trans4" : (s_next == 2 ? "stage_trans2" : "stage_trans1"), st);   // per target scale: bench.py prices each
    if ((s_next == 2 || s_next == 4) && (stage_quad_mask() & s_next)) {
        const int qtx = cdiv(Ws, 16);
        const int q_tiles = qtx * cdiv(Hs, 4);
        const int q_per = option(kOptXcdBands) ? cdiv(q_tiles, 8) : 0;
        dim3 qgrid(q_per ? q_per * 8 : q_tiles, B);
#define VFI_SQ(SPV, HP, NFV) \
    hipLaunchKernelGGL((stage_trans_quad_kernel<SPV, HP, NFV, 0>), qgrid, dim3(256), 0, st, Ppool, pack_stride, tasks, T, F, X, Hp, Wp, qtx, q_per, q_tiles)
#define VFI_SQ2(SPV, HP) \
    do { if (NF == 1) VFI_SQ(SPV, HP, 1); else VFI_SQ(SPV, HP, 2); } while (0)
        if (s_next == 4) {
            if (has_prev) VFI_SQ2(4, true); else VFI_SQ2(4, false);
        } else {
            if (has_prev) VFI_SQ2(2, true); else VFI_SQ2(2, false);
        }
#undef VFI_SQ2
#undef VFI_SQ
        VFI_CHECK_HIP(hipGetLastError());
        return 0;
    }
#define VFI_ST(SPV, HP, NFV) \
    hipLaunchKernelGGL((stage_trans_kernel<SPV, HP, NFV>), grid, dim3(256), 0, st, Ppool, pack_stride, tasks, T, F, X, Hp, Wp, tiles_x)
#define VFI_ST2(SPV, HP) \
    do { if (NF == 1) VFI_ST(SPV, HP, 1); else VFI_ST(SPV, HP, 2); } while (0)
    if (s_next == 4) {
        if (has_prev) VFI_ST2(4, true); else VFI_ST2(4, false);
    } else if (s_next == 2) {
        if (has_prev) VFI_ST2(2, true); else VFI_ST2(2, false);
    } else {
        if (has_prev) VFI_ST2(1, true); else VFI_ST2(1, false);
    }
#undef VFI_ST2
#undef VFI_ST
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Last block transition (block scales 2 -> 1) FUSED INTO the next block's first convolution (conv0.0: 3x3, stride 2,
// 20 -> 32 channels, LeakyReLU 0.2; rife_arch.py:237-249 on the input of :631-645).  The un-fused pair moves the
// full-resolution block input X through HBM twice — stage_trans<1> writes 24 channels x 1088x1920 x 4 B = 200 MB per frame
// and the convolution reads them back with its halo (profiles/r02_pmc_*: 3.66 GB written + 4.1 GB fetched per 16 frames,
// 1.67 + 1.36 ms, both at HBM speed).  Here a workgroup owns a 16x8 tile of the convolution's OUTPUT: it computes the
// 33x17-pixel patch of X that the tile needs straight into LDS — the same arithmetic as stage_trans_kernel<1> (flow update,
// both warps of image + features, timestep / mask / flow channels), pixels outside the image = the convolution's zero
// padding — and multiplies it on the fp32 matrix cores against the layer's weights, which stay in registers (27 K-steps x 4
// floats per lane) for all the tiles the persistent workgroup walks.  X never exists in HBM.
// The flow update cannot be in place any more (a tile's halo pixels belong to its neighbours, which may already have
// updated them): the new flow goes to a second buffer and the caller swaps.
// ---------------------------------------------------------------------------------------
constexpr int F0_TWO = 16, F0_THO = 8;                    // output tile (pixels of the stride-2 convolution)
constexpr int F0_TWI = 2 * F0_TWO + 1, F0_THI = 2 * F0_THO + 1;   // 33 x 17 input patch incl. the 1-pixel halo (top / left)
constexpr int F0_NPIX = F0_TWI * F0_THI;                  // 561
// LDS pixel stride = the 20 real channels (S/4 odd: conflict-free b128 reads).  The third 8-channel K chunk reads channels
// 16..23 of a pixel, i.e. 4 floats of the NEXT pixel for lanes 32-63: finite values that meet the zero weights of the padded
// input channels 20..23 (the last pixel's over-read lands in the weight image that follows the patch in LDS).
constexpr int F0_S = 20;
constexpr int F0_XF = F0_NPIX * F0_S;                     // floats of the patch
constexpr int F0_WF = 27 * 32 * 8;                        // floats of the layer's packed weights [tap][Cin/8][32][8]
constexpr int F0_THREADS = 512;
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// 8 waves: phase A has one patch pixel per thread (+ the 49 halo pixels on a second, overlapped round); in phase B wave w
// multiplies sub-tile (w & 3) over one half of the 27 K-steps (w >> 2), the halves are added through LDS.
__global__ __launch_bounds__(F0_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void trans1_conv0a_kernel(const float* __restrict__ Ppool, size_t pack_stride, RifeTasks tasks,
                                                                   const float* __restrict__ T, const float* __restrict__ Fin,
                                                                   float* __restrict__ Fout, const float* __restrict__ wpk,
                                                                   const float* __restrict__ bias, float* __restrict__ A0, int Hp,
                                                                   int Wp, int n_tiles, int tiles_x, int tiles_y, float slope) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float lds[];   // F0_XF floats of patch, then F0_WF floats of weights
    float* wl = lds + F0_XF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int Ho = Hp / 2, Wo = Wp / 2;
    const int Hi = Hp / 2, Wi = Wp / 2;          // resolution of the pixel-shuffled T of the scale-2 block
    for (int i = tid; i < F0_WF / 4; i += F0_THREADS) ((float4*)wl)[i] = ((const float4*)wpk)[i];   // once per persistent workgroup
    const float bs = bias[l31];
    // A fragment base: sub-tile (2 x 2 sub-tiles of 8 x 4 pixels), tap origin (-1, -1) folded in
    const int sub = wave & 3, khalf = wave >> 2;
    const int sx = sub & 1, sy = sub >> 1;
    const int oyl = sy * 4 + (l31 >> 3), oxl = sx * 8 + (l31 & 7);
    const int abase = ((2 * oyl) * F0_TWI + 2 * oxl) * F0_S + half * 4;
    const int bbase = l31 * 8 + half * 4;
    const size_t hi_off = (size_t)Hp * Wp * 4;
    const WarpGeo g = make_warp_geo(Wp, Hp);
    const int tiles_per_img = tiles_x * tiles_y;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int b = tile / tiles_per_img;
        const int trem = tile - b * tiles_per_img;
        const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
        const int Y0 = ty * F0_THO, X0 = tx * F0_TWO;
        const int iy0 = 2 * Y0 - 1, ix0 = 2 * X0 - 1;
        const float* Tb = T + (size_t)b * Hi * Wi * 8;
        const float* P0 = Ppool + (size_t)tasks.slot0[b] * pack_stride;
        const float* P1 = Ppool + (size_t)tasks.slot1[b] * pack_stride;
        const float tstep = tasks.t[b];
        // ---- phase A: the X patch (stage_trans_kernel<1, true, 1> per pixel)
        // Raised issue priority: the CU's other workgroup is usually in phase B, and an fp32 MFMA keeps every VALU instruction of its
        // SIMD out for 64 cycles — at equal priority this phase's address arithmetic (hence its loads) only trickles out between them.
        // With priority the loads leave early and the MFMAs fill the wait (3.48 -> 3.32 ms per launch together with the fragment
        // addressing below; profiles/r04_trans1_conv0a_phases.txt).
        __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const int p = tid + F0_THREADS * it;
            if (p >= F0_NPIX) break;
            const int py = p / F0_TWI, px = p - py * F0_TWI;
            const int Y = iy0 + py, X = ix0 + px;
            float r[20];
#pragma unroll
            for (int c = 0; c < 20; ++c) r[c] = 0.f;
            if (Y >= 0 && X >= 0 && Y < Hp && X < Wp) {
                const Bil by = bil_index(Y, 0.5f, Hi), bx = bil_index(X, 0.5f, Wi);
                const TVal v = t_bilerp(t_read(Tb, Hi, Wi, by.i0, bx.i0), t_read(Tb, Hi, Wi, by.i0, bx.i1), t_read(Tb, Hi, Wi, by.i1, bx.i0),
                                        t_read(Tb, Hi, Wi, by.i1, bx.i1), by.w0, by.w1, bx.w0, bx.w1);
                const size_t pb = (size_t)b * Hp * Wp + (size_t)Y * Wp + X;
                const float4 o4 = ((const float4*)Fin)[pb];
                const float4 f = make_float4(o4.x + v.f.x * 2.0f, o4.y + v.f.y * 2.0f, o4.z + v.f.z * 2.0f, o4.w + v.f.w * 2.0f);
                if (py >= 1 && px >= 1) ((float4*)Fout)[pb] = f;        // this tile owns rows 2*Y0.. and columns 2*X0..
                const Tap4 t0 = warp_taps(g, X, Y, f.x, f.y);
                const Tap4 t1 = warp_taps(g, X, Y, f.z, f.w);
                float4 a_lo, b_lo, a_hi[1], b_hi[1];
                sample_pack<1>(P0, hi_off, t0, a_lo, a_hi);
                sample_pack<1>(P1, hi_off, t1, b_lo, b_hi);
                cat_inputs<1>(r, a_lo, b_lo, a_hi, b_hi, tstep);   // channels 0..14: img0, img1, feat0, feat1, timestep
                r[15] = v.m;
                r[16] = f.x, r[17] = f.y, r[18] = f.z, r[19] = f.w;     // interpolate(flow, 1/1) * 1/1
            }
            float* d = &lds[p * F0_S];
#pragma unroll
            for (int q = 0; q < 5; ++q) *(float4*)(d + 4 * q) = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        // ---- phase B: 128 output pixels x 32 channels, K = 9 taps x 24 channels; this wave: K-steps [14 khalf, 14 khalf + 14)
        f32x16_t acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        {
            // Fragment addresses = ONE opaque byte base per operand + compile-time offsets (the ds_read_b128 offset field).  Left to
            // itself hipcc hoists the 27 + 27 per-step addresses out of the tile loop and keeps them in 28 VGPRs through phase A
            // (128-register budget: 2 spilled; now 102, none).  khalf is wave-uniform: each half is its own straight-line code.
            int ab = abase * 4, bb = bbase * 4;
            asm volatile("" : "+v"(ab), "+v"(bb));
            const char* const ap = (const char*)lds + ab;
            const char* const bp = (const char*)wl + bb;
            auto steps = [&](auto KH) {
                constexpr int kh = decltype(KH)::value;
#pragma unroll
                for (int k = 0; k < 14; ++k) {
                    const int st = kh * 14 + k;
                    if (st < 27) {
                        const int t = st / 3, c8 = st - 3 * t;
                        const int toff = ((t / 3) * F0_TWI + (t % 3)) * F0_S;
                        const f32x4_t av = *(const f32x4_t*)(ap + (toff + c8 * 8) * 4);
                        const f32x4_t bv = *(const f32x4_t*)(bp + st * 1024);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[j], acc, 0, 0, 0);
                    }
                    if (k & 1) __builtin_amdgcn_sched_barrier(0);    // at most two steps' fragments in flight
                }
            };
            if (khalf) steps(std::integral_constant<int, 1>{});
            else steps(std::integral_constant<int, 0>{});
        }
        __syncthreads();   // the patch is consumed: its LDS now carries the upper K-half's partial sums
        f32x16_t* red = (f32x16_t*)lds;
        if (khalf) red[sub * 64 + lane] = acc;
        __syncthreads();
        if (!khalf) {
            const f32x16_t other = red[sub * 64 + lane];
            // ---- epilogue: + bias, LeakyReLU, NHWC store (32 lanes = 128 contiguous bytes per pixel)
            int ot = tid;                     // lane part of the store address re-derived here: hoisted out of the tile loop it is one
            asm volatile("" : "+v"(ot));      // more 64-bit value live through phase A
            const int oy0 = Y0 + sy * 4, ox0 = X0 + sx * 8 + 4 * ((ot >> 5) & 1);
            float* ob = A0 + ((size_t)(b * Ho + oy0) * Wo + ox0) * 32 + (ot & 31);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float v = (acc[i] + other[i]) + bs;
                ob[((size_t)(i >> 2) * Wo + (i & 3)) * 32] = fmaxf(v, v * slope);
            }
        }
        __syncthreads();   // the partial sums are consumed; the next tile may overwrite the patch
    }
#endif
}

int trans1_conv0a_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* T, const float* Fin,
                         float* Fout, const float* wpk, const float* bias, float* A0, int Hp, int Wp, float slope, hipStream_t st) {
    VFI_REQUIRE(Hp % 32 == 0 && Wp % 32 == 0 && slope >= 0.f && slope <= 1.f, "trans1_conv0a: bad geometry %dx%d / slope %g", Hp, Wp, slope);
    const int tiles_x = (Wp / 2) / F0_TWO, tiles_y = (Hp / 2) / F0_THO;
    const int n_tiles = B * tiles_x * tiles_y;
    static std::atomic<int> cus_of[kMaxDevices];
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "trans1_conv0a: device index %d out of range", dev);
    int cus = cus_of[dev].load(std::memory_order_relaxed);
    if (!cus) {
        hipDeviceProp_t p;
        VFI_CHECK_HIP(hipGetDeviceProperties(&p, dev));
        cus = p.multiProcessorCount;
        cus_of[dev].store(cus, std::memory_order_relaxed);
    }
    // 72.5 KB of LDS (patch 44.9 + weights 27.6) and 512 threads: two workgroups per CU, each walks its share of the tiles
    static bool attr_set[kMaxDevices] = {};
    if (!attr_set[dev]) {
        VFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trans1_conv0a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (F0_XF + F0_WF) * 4));
        attr_set[dev] = true;
    }
    const int grid = std::min(n_tiles, 2 * launch_cus(cus));
    TraceScope ts("trans1_conv0a", st);
    hipLaunchKernelGGL(trans1_conv0a_kernel, dim3(grid), dim3(F0_THREADS), (F0_XF + F0_WF) * 4, st, Ppool, pack_stride, tasks, T, Fin, Fout, wpk, bias, A0, Hp, Wp,
                       n_tiles, tiles_x, tiles_y, slope);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// fused block transition for arch 4.26: as stage_trans_kernel, plus the 8 feature channels the block returns
// (T planes 1..3 = mask, g0..g7) which the next block takes un-warped, between mask and flow (rife_arch.py:555-583).
// Block scales 2*SP -> SP with SP in {8, 4, 2, 1} (the first transition of 4.26 is 16 -> 8).
// ---------------------------------------------------------------------------------------
struct TXVal {
    float4 f, p1, p2;   // flow delta | mask, g0, g1, g2 | g3..g6
    float g7;
};
__device__ static inline TXVal tx_read(const float* __restrict__ Tb, int Hs, int Ws, int Yt, int Xt) {
    const size_t p = (size_t)Yt * Ws + Xt, ps = (size_t)Hs * Ws;
    TXVal v;
    v.f = ((const float4*)Tb)[p];
    v.p1 = ((const float4*)Tb)[ps + p];
    v.p2 = ((const float4*)Tb)[2 * ps + p];
    v.g7 = Tb[(3 * ps + p) * 4];
    return v;
}

template <int SP, bool HAS_PREV>
__global__ __launch_bounds__(256) void stage_trans_x_kernel(const float* __restrict__ Ppool, size_t pack_stride,
                                                            RifeTasks tasks, const float* __restrict__ T,
                                                            float* __restrict__ F, float* __restrict__ Xo, int Hp,
                                                            int Wp, int tiles_x) {
    constexpr int SI = 2 * SP;
    constexpr int NP = SP == 1 ? 1 : 2;
    constexpr int OFF = SP == 1 ? 0 : SP / 2 - 1;
    const int Hs = Hp / SP, Ws = Wp / SP;
    const int b = blockIdx.y;
    const int tile_y = blockIdx.x / tiles_x, tile_x = blockIdx.x - tile_y * tiles_x;
    const int xl = tile_x * 16 + (threadIdx.x & 15), yl = tile_y * 16 + (threadIdx.x >> 4);
    if (xl >= Ws || yl >= Hs) return;
    const int Hi = Hp / SI, Wi = Wp / SI;
    const float* Tb = T + (size_t)b * Hi * Wi * 16;
    const float rs = 1.0f / (float)SI;
    const Bil by0 = bil_index(yl * SP, rs, Hi), bx0 = bil_index(xl * SP, rs, Wi);
    const TXVal t00 = tx_read(Tb, Hi, Wi, by0.i0, bx0.i0), t01 = tx_read(Tb, Hi, Wi, by0.i0, bx0.i1);
    const TXVal t10 = tx_read(Tb, Hi, Wi, by0.i1, bx0.i0), t11 = tx_read(Tb, Hi, Wi, by0.i1, bx0.i1);
#define VFI_BL(A, B, C, D) \
    __fadd_rn(__fmul_rn(wy0, __fadd_rn(__fmul_rn(wx0, A), __fmul_rn(wx1, B))), __fmul_rn(wy1, __fadd_rn(__fmul_rn(wx0, C), __fmul_rn(wx1, D))))
    // ---- phase 1: flow (+)= up(T)*SI for every pixel of the cell; keep flow, mask and features of the centre pixels
    float4 fc[NP * NP];
    float mc[NP * NP], gc[NP * NP][8];
#pragma unroll 1
    for (int dy = 0; dy < SP; ++dy) {
        const Bil wyb = bil_index(yl * SP + dy, rs, Hi);
        const float wy0 = wyb.w0, wy1 = wyb.w1;
#pragma unroll
        for (int dx = 0; dx < SP; ++dx) {
            const Bil wxb = bil_index(xl * SP + dx, rs, Wi);
            const float wx0 = wxb.w0, wx1 = wxb.w1;
            const size_t pb = (size_t)b * Hp * Wp + (size_t)(yl * SP + dy) * Wp + xl * SP + dx;
            const float fs = (float)SI;
            float4 f = make_float4(VFI_BL(t00.f.x, t01.f.x, t10.f.x, t11.f.x) * fs, VFI_BL(t00.f.y, t01.f.y, t10.f.y, t11.f.y) * fs,
                                   VFI_BL(t00.f.z, t01.f.z, t10.f.z, t11.f.z) * fs, VFI_BL(t00.f.w, t01.f.w, t10.f.w, t11.f.w) * fs);
            if (HAS_PREV) {
                const float4 o = ((const float4*)F)[pb];
                f = make_float4(o.x + f.x, o.y + f.y, o.z + f.z, o.w + f.w);
            }
            ((float4*)F)[pb] = f;
            const int cy = dy - OFF, cx = dx - OFF;
            if (cy >= 0 && cy < NP && cx >= 0 && cx < NP) {
                const float m = VFI_BL(t00.p1.x, t01.p1.x, t10.p1.x, t11.p1.x);
                const float g[8] = {VFI_BL(t00.p1.y, t01.p1.y, t10.p1.y, t11.p1.y), VFI_BL(t00.p1.z, t01.p1.z, t10.p1.z, t11.p1.z),
                                    VFI_BL(t00.p1.w, t01.p1.w, t10.p1.w, t11.p1.w), VFI_BL(t00.p2.x, t01.p2.x, t10.p2.x, t11.p2.x),
                                    VFI_BL(t00.p2.y, t01.p2.y, t10.p2.y, t11.p2.y), VFI_BL(t00.p2.z, t01.p2.z, t10.p2.z, t11.p2.z),
                                    VFI_BL(t00.p2.w, t01.p2.w, t10.p2.w, t11.p2.w), VFI_BL(t00.g7, t01.g7, t10.g7, t11.g7)};
#pragma unroll
                for (int q = 0; q < NP * NP; ++q)   // static indices only (cy, cx depend on the rolled dy)
                    if (q == cy * NP + cx) {
                        fc[q] = f;
                        mc[q] = m;
#pragma unroll
                        for (int j = 0; j < 8; ++j) gc[q][j] = g[j];
                    }
            }
        }
    }
#undef VFI_BL
    // ---- phase 2: warp both frame packs at the centre pixels, down-resize (weights 1/2), write X
    const float* P0 = Ppool + (size_t)tasks.slot0[b] * pack_stride;
    const float* P1 = Ppool + (size_t)tasks.slot1[b] * pack_stride;
    const size_t hi_off = (size_t)Hp * Wp * 4;
    const float tstep = tasks.t[b];
    const WarpGeo g = make_warp_geo(Wp, Hp);
    constexpr int NC = 28, NR = 32;   // img 6, features 8, timestep, mask, carried 8, flow 4 (+4 zero)
    float r[NR], row[NC], o[NC];
#pragma unroll
    for (int c = 0; c < NR; ++c) r[c] = 0.f;
#pragma unroll 1
    for (int k = 0; k < NP * NP; ++k) {
        const int dy = k / NP, dx = k % NP;
        float4 f = fc[0];
        float m = mc[0];
        float gg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gg[j] = gc[0][j];
#pragma unroll
        for (int q = 1; q < NP * NP; ++q) {
            if (k == q) {
                f = fc[q];
                m = mc[q];
#pragma unroll
                for (int j = 0; j < 8; ++j) gg[j] = gc[q][j];
            }
        }
        const int Y = yl * SP + OFF + dy, X = xl * SP + OFF + dx;
        const Tap4 t0 = warp_taps(g, X, Y, f.x, f.y);
        const Tap4 t1 = warp_taps(g, X, Y, f.z, f.w);
        float4 a_lo, b_lo, a_hi[1], b_hi[1];
        sample_pack<1>(P0, hi_off, t0, a_lo, a_hi);
        sample_pack<1>(P1, hi_off, t1, b_lo, b_hi);
        cat_inputs<1>(o, a_lo, b_lo, a_hi, b_hi, tstep);
        o[15] = m;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[16 + j] = gg[j];
        o[24] = f.x; o[25] = f.y; o[26] = f.z; o[27] = f.w;
#pragma unroll
        for (int c = 0; c < NC; ++c) row[c] = dx == 0 ? o[c] : __fadd_rn(0.5f * row[c], 0.5f * o[c]);
        if (dx == NP - 1) {
#pragma unroll
            for (int c = 0; c < NC; ++c) r[c] = dy == 0 ? row[c] : __fadd_rn(0.5f * r[c], 0.5f * row[c]);
        }
    }
    const float inv_s = 1.0f / (float)SP;
#pragma unroll
    for (int c = NC - 4; c < NC; ++c) r[c] = r[c] * inv_s;
    float4* op = (float4*)Xo + (size_t)b * (NR / 4) * Hs * Ws + (size_t)yl * Ws + xl;
#pragma unroll
    for (int q = 0; q < NR / 4; ++q) op[(size_t)q * Hs * Ws] = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
}

int stage_trans_x_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* T, float* F,
                         float* X, int Hp, int Wp, int s_prev, int s_next, bool has_prev, hipStream_t st) {
    VFI_REQUIRE(s_prev == 2 * s_next && (s_next == 8 || s_next == 4 || s_next == 2 || s_next == 1),
                "stage_trans_x: scales %d -> %d not on the fused path", s_prev, s_next);
    const int Hs = Hp / s_next, Ws = Wp / s_next;
    const int tiles_x = cdiv(Ws, 16), tiles_y = cdiv(Hs, 16);
    dim3 grid(tiles_x * tiles_y, B);
    TraceScope ts(s_next == 8 ? "stage_transx8" : (s_next == 4 ? "stage_transx4" : (s_next == 2 ? "stage_transx2" : "stage_transx1")), st);
    if (s_next >= 2 && (stage_quad_mask() & s_next)) {
        const int qtx = cdiv(Ws, 16);
        const int q_tiles = qtx * cdiv(Hs, 4);
        const int q_per = option(kOptXcdBands) ? cdiv(q_tiles, 8) : 0;
        dim3 qgrid(q_per ? q_per * 8 : q_tiles, B);
#define VFI_SQ(SPV, HP) \
    hipLaunchKernelGGL((stage_trans_quad_kernel<SPV, HP, 1, 8>), qgrid, dim3(256), 0, st, Ppool, pack_stride, tasks, T, F, X, Hp, Wp, qtx, q_per, q_tiles)
#define VFI_SQ2(SPV) \
    do { if (has_prev) VFI_SQ(SPV, true); else VFI_SQ(SPV, false); } while (0)
        if (s_next == 8) VFI_SQ2(8);
        else if (s_next == 4) VFI_SQ2(4);
        else VFI_SQ2(2);
#undef VFI_SQ2
#undef VFI_SQ
        VFI_CHECK_HIP(hipGetLastError());
        return 0;
    }
#define VFI_ST(SPV, HP) \
    hipLaunchKernelGGL((stage_trans_x_kernel<SPV, HP>), grid, dim3(256), 0, st, Ppool, pack_stride, tasks, T, F, X, Hp, Wp, tiles_x)
#define VFI_ST2(SPV) \
    do { if (has_prev) VFI_ST(SPV, true); else VFI_ST(SPV, false); } while (0)
    if (s_next == 8) VFI_ST2(8);
    else if (s_next == 4) VFI_ST2(4);
    else if (s_next == 2) VFI_ST2(2);
    else VFI_ST2(1);
#undef VFI_ST2
#undef VFI_ST
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// last stage fused with the output: flow += tmp*s, mask = tmp[4], warp both images,
// sigmoid blend, crop to HxW, node-level clamp(0,1)            rife_arch.py:703-704,721-723,732
//                                                               rife/__init__.py:207
// ---------------------------------------------------------------------------------------
__global__ void final_blend_kernel(const float* __restrict__ Ppool, size_t pack_stride, RifeTasks tasks,
                                   const float* __restrict__ T, const float* __restrict__ F, float* __restrict__ out,
                                   float* __restrict__ Fdbg, int H, int W, int Hp, int Wp, int s, int tp, int xcd_per) {
    // xcd_per > 0: XCD-aware order — workgroup x of a frame runs on XCD x % 8 (the launch's x extent is a multiple of 8), so XCD k takes the
    // contiguous run of 256-pixel segments [k * xcd_per, (k + 1) * xcd_per): the two pack rows a bilinear tap pair touches are then read
    // through ONE L2 instead of two (r6 PMC: 1.345x the algorithmic bytes with the plain order)
    const int seg = xcd_per > 0 ? ((int)blockIdx.x & 7) * xcd_per + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int idx = seg * blockDim.x + threadIdx.x;
    if (idx >= H * W || (xcd_per > 0 && ((int)blockIdx.x >> 3) >= xcd_per)) return;
    const int b = blockIdx.y;
    const int X = idx % W, Y = idx / W;
    const int Hs = Hp / s, Ws = Wp / s;
    const float* Tb = T + (size_t)b * Hs * Ws * 4 * tp;
    const TVal tv = t_upsample(Tb, Hs, Ws, s, Y, X);
    const size_t pb = (size_t)b * Hp * Wp + (size_t)Y * Wp + X;
    const float fs = (float)s;
    const float4 o = ((const float4*)F)[pb];
    const float4 f = make_float4(o.x + tv.f.x * fs, o.y + tv.f.y * fs, o.z + tv.f.z * fs, o.w + tv.f.w * fs);
    if (Fdbg) ((float4*)Fdbg)[pb] = f;
    const WarpGeo g = make_warp_geo(Wp, Hp);
    const Tap4 t0 = warp_taps(g, X, Y, f.x, f.y);
    const Tap4 t1 = warp_taps(g, X, Y, f.z, f.w);
    float4 a, bb;
    sample_pack<0>(Ppool + (size_t)tasks.slot0[b] * pack_stride, 0, t0, a, nullptr);
    sample_pack<0>(Ppool + (size_t)tasks.slot1[b] * pack_stride, 0, t1, bb, nullptr);
    const float m = 1.0f / (1.0f + expf(-tv.m));
    const float om = 1.0f - m;
    float* op = out + ((size_t)b * H * W + idx) * 3;
    op[0] = fminf(fmaxf(a.x * m + bb.x * om, 0.f), 1.f);
    op[1] = fminf(fmaxf(a.y * m + bb.y * om, 0.f), 1.f);
    op[2] = fminf(fmaxf(a.z * m + bb.z * om, 0.f), 1.f);
}

int final_blend_launch(const float* Ppool, size_t pack_stride, const RifeTasks& tasks, int B, const float* T,
                       const float* F, float* out, float* Fdbg, int H, int W, int Hp, int Wp, int s, int tp,
                       hipStream_t st) {
    const int segs = cdiv(H * W, 256);
    const int xcd_per = option(kOptXcdBands) ? cdiv(segs, 8) : 0;
    dim3 grid(xcd_per ? xcd_per * 8 : segs, B);
    TraceScope ts("final_blend", st);
    hipLaunchKernelGGL(final_blend_kernel, grid, dim3(256), 0, st, Ppool, pack_stride, tasks, T, F, out, Fdbg, H, W,
                       Hp, Wp, s, tp, xcd_per);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Fractional block scales (node widget scale_factor 2 / 4 -> block scales 0.5 / 0.25): the block runs ABOVE the
// frame resolution.  IFBlock.forward (rife_arch.py:237-276) then up-samples its input by u = 1/scale
// (flow channels additionally * u) and down-samples its output by scale (flow * scale).  Both resizes are plain
// bilinear align_corners=False; they are done on the assembled planar4 tensors around the unchanged block kernels:
//   stage_in(scale 1) -> X1 --planar4_up--> X (u*Hp x u*Wp) -> convs -> T (u*Hp x u*Wp) --t_down--> T1 (Hp x Wp, scale 1)
// ---------------------------------------------------------------------------------------
__global__ void planar4_up_kernel(const float* __restrict__ X1, float* __restrict__ Xo, int Hp, int Wp, int u, int nplanes,
                                  int flow_plane) {
    const int Hs = Hp * u, Ws = Wp * u;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Hs * Ws) return;
    const int pl = blockIdx.y, b = blockIdx.z;
    const int X = idx % Ws, Y = idx / Ws;
    const float rs = 1.0f / (float)u;
    const Bil by = bil_index(Y, rs, Hp), bx = bil_index(X, rs, Wp);
    const float4* src = (const float4*)X1 + ((size_t)b * nplanes + pl) * Hp * Wp;
    const float4 a = src[(size_t)by.i0 * Wp + bx.i0], bb = src[(size_t)by.i0 * Wp + bx.i1];
    const float4 c = src[(size_t)by.i1 * Wp + bx.i0], d = src[(size_t)by.i1 * Wp + bx.i1];
    const float wy0 = by.w0, wy1 = by.w1, wx0 = bx.w0, wx1 = bx.w1;
#define VFI_BL(A, B, C, D) \
    __fadd_rn(__fmul_rn(wy0, __fadd_rn(__fmul_rn(wx0, A), __fmul_rn(wx1, B))), __fmul_rn(wy1, __fadd_rn(__fmul_rn(wx0, C), __fmul_rn(wx1, D))))
    float4 r;
    r.x = VFI_BL(a.x, bb.x, c.x, d.x);
    r.y = VFI_BL(a.y, bb.y, c.y, d.y);
    r.z = VFI_BL(a.z, bb.z, c.z, d.z);
    r.w = VFI_BL(a.w, bb.w, c.w, d.w);
#undef VFI_BL
    if (pl == flow_plane) {  // interpolate(flow, 1/scale) * 1.0 / scale: the factor is a power of two, exact
        const float fu = (float)u;
        r = make_float4(r.x * fu, r.y * fu, r.z * fu, r.w * fu);
    }
    ((float4*)Xo)[((size_t)b * nplanes + pl) * Hs * Ws + idx] = r;
}

int planar4_up_launch(const float* X1, float* X, int B, int Hp, int Wp, int u, int CX, int flow_plane, hipStream_t st) {
    VFI_REQUIRE(u == 2 || u == 4, "planar4_up: factor %d", u);
    dim3 grid(cdiv((long)Hp * u * Wp * u, 256), CX / 4, B);
    TraceScope ts("stage_up", st);
    hipLaunchKernelGGL(planar4_up_kernel, grid, dim3(256), 0, st, X1, X, Hp, Wp, u, CX / 4, flow_plane);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// T [B][tp][u*Hp][u*Wp][4] -> T1 [B][tp][Hp][Wp][4]: interpolate(tmp, scale_factor=1/u); plane 0 (flow) * (1/u)
__global__ void t_down_kernel(const float* __restrict__ T, float* __restrict__ T1, int Hp, int Wp, int u, int tp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hp * Wp) return;
    const int pl = blockIdx.y, b = blockIdx.z;
    const int X = idx % Wp, Y = idx / Wp;
    const int Hs = Hp * u, Ws = Wp * u;
    const float4* src = (const float4*)T + ((size_t)b * tp + pl) * Hs * Ws;
    const Bil by = bil_index(Y, (float)u, Hs), bx = bil_index(X, (float)u, Ws);
    const float4 a = src[(size_t)by.i0 * Ws + bx.i0], bb = src[(size_t)by.i0 * Ws + bx.i1];
    const float4 c = src[(size_t)by.i1 * Ws + bx.i0], d = src[(size_t)by.i1 * Ws + bx.i1];
    const float wy0 = by.w0, wy1 = by.w1, wx0 = bx.w0, wx1 = bx.w1;
#define VFI_BL(A, B, C, D) \
    __fadd_rn(__fmul_rn(wy0, __fadd_rn(__fmul_rn(wx0, A), __fmul_rn(wx1, B))), __fmul_rn(wy1, __fadd_rn(__fmul_rn(wx0, C), __fmul_rn(wx1, D))))
    float4 r;
    r.x = VFI_BL(a.x, bb.x, c.x, d.x);
    r.y = VFI_BL(a.y, bb.y, c.y, d.y);
    r.z = VFI_BL(a.z, bb.z, c.z, d.z);
    r.w = VFI_BL(a.w, bb.w, c.w, d.w);
#undef VFI_BL
    if (pl == 0) {
        const float sc = 1.0f / (float)u;
        r = make_float4(r.x * sc, r.y * sc, r.z * sc, r.w * sc);
    }
    ((float4*)T1)[((size_t)b * tp + pl) * Hp * Wp + idx] = r;
}

int t_down_launch(const float* T, float* T1, int B, int Hp, int Wp, int u, int tp, hipStream_t st) {
    VFI_REQUIRE(u == 2 || u == 4, "t_down: factor %d", u);
    dim3 grid(cdiv(Hp * Wp, 256), tp, B);
    TraceScope ts("t_down", st);
    hipLaunchKernelGGL(t_down_kernel, grid, dim3(256), 0, st, T, T1, Hp, Wp, u, tp);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// arch 4.26: the block also returns 8 feature channels (tmp[:, 5:13] after interpolate(tmp, scale), rife_arch.py:267-273)
// that are fed to the next block.  T has 4 planes: (flow 4 | mask, f0, f1, f2 | f3..f6 | f7, -, -, -);
// FEAT [B][2][Hp][Wp][4] = those 8 channels up-resized by s.
__global__ void feat_up_kernel(const float* __restrict__ T, float* __restrict__ FEAT, int Hp, int Wp, int s) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hp * Wp) return;
    const int b = blockIdx.y;
    const int X = idx % Wp, Y = idx / Wp;
    const int Hs = Hp / s, Ws = Wp / s;
    const float4* src = (const float4*)T + (size_t)b * 4 * Hs * Ws;
    const size_t ps = (size_t)Hs * Ws;
    const float rs = 1.0f / (float)s;
    const Bil by = s == 1 ? Bil{Y, Y, 1.f, 0.f} : bil_index(Y, rs, Hs);
    const Bil bx = s == 1 ? Bil{X, X, 1.f, 0.f} : bil_index(X, rs, Ws);
    const size_t o00 = (size_t)by.i0 * Ws + bx.i0, o01 = (size_t)by.i0 * Ws + bx.i1;
    const size_t o10 = (size_t)by.i1 * Ws + bx.i0, o11 = (size_t)by.i1 * Ws + bx.i1;
    const float wy0 = by.w0, wy1 = by.w1, wx0 = bx.w0, wx1 = bx.w1;
    float v[12];
#define VFI_BL(A, B, C, D) \
    (s == 1 ? (A) : __fadd_rn(__fmul_rn(wy0, __fadd_rn(__fmul_rn(wx0, A), __fmul_rn(wx1, B))), __fmul_rn(wy1, __fadd_rn(__fmul_rn(wx0, C), __fmul_rn(wx1, D)))))
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const float4 a = src[(1 + pl) * ps + o00], bb = src[(1 + pl) * ps + o01], c = src[(1 + pl) * ps + o10], d = src[(1 + pl) * ps + o11];
        v[4 * pl] = VFI_BL(a.x, bb.x, c.x, d.x);
        v[4 * pl + 1] = VFI_BL(a.y, bb.y, c.y, d.y);
        v[4 * pl + 2] = VFI_BL(a.z, bb.z, c.z, d.z);
        v[4 * pl + 3] = VFI_BL(a.w, bb.w, c.w, d.w);
    }
#undef VFI_BL
    float4* o = (float4*)FEAT + (size_t)b * 2 * Hp * Wp + idx;
    o[0] = make_float4(v[1], v[2], v[3], v[4]);
    o[(size_t)Hp * Wp] = make_float4(v[5], v[6], v[7], v[8]);
}

int feat_up_launch(const float* T, float* FEAT, int B, int Hp, int Wp, int s, hipStream_t st) {
    dim3 grid(cdiv(Hp * Wp, 256), B);
    TraceScope ts("feat_up", st);
    hipLaunchKernelGGL(feat_up_kernel, grid, dim3(256), 0, st, T, FEAT, Hp, Wp, s);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// deconv+PixelShuffle result in plain NHWC [N,4H,4W,C4] from the planar4 T layout (test entry, C4 <= 6)
__global__ void t_to_nhwc_kernel(const float* __restrict__ T, float* __restrict__ out, int N, int Hq, int Wq, int C4) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Hs = Hq * 4, Ws = Wq * 4;
    if (idx >= (long)N * Hs * Ws) return;
    const long p = idx % ((long)Hs * Ws);
    const int n = idx / ((long)Hs * Ws);
    const float* Tb = T + (size_t)n * Hs * Ws * 8;
    for (int c = 0; c < C4; ++c) out[idx * C4 + c] = Tb[((size_t)(c >> 2) * Hs * Ws + p) * 4 + (c & 3)];
}
int t_to_nhwc_launch(const float* T, float* out, int N, int Hq, int Wq, int C4, hipStream_t st) {
    const long total = (long)N * Hq * 4 * Wq * 4;
    hipLaunchKernelGGL(t_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, T, out, N, Hq, Wq, C4);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vfi
