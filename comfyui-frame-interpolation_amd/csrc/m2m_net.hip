// M2M-specific element-wise / gather kernels behind the C ABI (the convolutions, cost volume and splat are the
// generic entry points of gen_ops.hip / m2m_ops.hip; m2m.py strings them together).
//
// Reference semantics restated (vfi_models/m2m/M2M_arch.py): backwarp (grid_sample bilinear, zeros padding,
// align_corners=True) :24-92; input padding + joint normalisation :903-934; EncDec squeeze-excite "cube" :679-704,
// :786-795; photometric metric, splat inputs, forwarp_mframe_mask and the hole fill :551-581, :960-1033.
//
// Layouts: NHWC fp32 channel windows (`*_cs` = floats per pixel of the underlying tensor).  Tensors that hold the
// two directions of one frame pair are batches of 2 (image 0 = forward / frame 0, image 1 = backward / frame 1);
// "swap" arguments read the partner image (n ^ 1) — the reference's separate calls on (a, b) and (b, a).
#include <cmath>
#include <cstdio>

#include "../../include/vfi_hip.h"
#include "vfi_common.h"
#include "m2m_warp.h"

namespace vfi {

static unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

// ---- replicate padding + joint statistics ------------------------------------------------------------------
// stats over the PADDED images (the reference pads first, M2M_arch.py:903-913, then takes mean/std :915-931).
constexpr int ST_BLOCKS = 512;
__global__ __launch_bounds__(256) void m2m_stats_partial(const float* __restrict__ f0, const float* __restrict__ f1, int C,
                                                         int H, int W, int Hp, int Wp, double* __restrict__ part) {
    __shared__ double red[4][256];
    double s[4] = {0, 0, 0, 0};  // sum0, sumsq0, sum1, sumsq1
    const long total = (long)Hp * Wp;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
        const int y = min((int)(p / Wp), H - 1), x = min((int)(p % Wp), W - 1);
        const float* a = f0 + ((size_t)y * W + x) * C;
        const float* b = f1 + ((size_t)y * W + x) * C;
        for (int c = 0; c < 3; ++c) {
            const double va = a[c], vb = b[c];
            s[0] += va;
            s[1] += va * va;
            s[2] += vb;
            s[3] += vb * vb;
        }
    }
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 4) part[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}
// stats[0] = mean_, stats[1] = std_ + 1e-7 (the divisor / multiplier the reference uses everywhere)
// (one workgroup of 256: thread i sums blocks i, i + 256, ... in ascending order, then a fixed tree — deterministic; a single thread
// walking the 512 partials took 68 us of a 7.7 ms prepare)
__global__ __launch_bounds__(256) void m2m_stats_final(const double* __restrict__ part, int nblocks, double count, float* __restrict__ stats) {
    __shared__ double red[4][256];
    double s[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < nblocks; b += 256)
        for (int k = 0; k < 4; ++k) s[k] += part[(size_t)b * 4 + k];
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    for (int k = 0; k < 4; ++k) s[k] = red[k][0];
    const float m0 = (float)(s[0] / count), m1 = (float)(s[2] / count);
    const float mean = __fdiv_rn(__fadd_rn(m0, m1), 2.0f);
    const float v0 = (float)(s[1] / count - (s[0] / count) * (s[0] / count));
    const float v1 = (float)(s[3] / count - (s[2] / count) * (s[2] / count));
    const float d0 = __fsub_rn(mean, m0), d1 = __fsub_rn(mean, m1);
    const float var = __fdiv_rn(__fadd_rn(__fadd_rn(v0, __fmul_rn(d0, d0)), __fadd_rn(v1, __fmul_rn(d1, d1))), 2.0f);
    stats[0] = mean;
    stats[1] = __fadd_rn(sqrtf(var), 0.0000001f);
}
// D0[n, y, x, coff + c] = (frame_n[clamp(y), clamp(x), c] - mean) / (std + 1e-7)
__global__ void m2m_normalize_kernel(const float* __restrict__ f0, const float* __restrict__ f1, int C, int H, int W, int Hp,
                                     int Wp, const float* __restrict__ stats, float* __restrict__ out, int out_cs, int coff) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2L * Hp * Wp) return;
    const int n = idx / ((long)Hp * Wp);
    const long p = idx - (long)n * Hp * Wp;
    const int y = min((int)(p / Wp), H - 1), x = min((int)(p % Wp), W - 1);
    const float* src = (n ? f1 : f0) + ((size_t)y * W + x) * C;
    const float mean = stats[0], sd = stats[1];
    float* o = out + (size_t)idx * out_cs + coff;
    for (int c = 0; c < 3; ++c) o[c] = __fdiv_rn(__fsub_rn(src[c], mean), sd);
}

template <bool VEC>
__global__ void warp_m2m_kernel(const float* __restrict__ in, int in_cs, int in_swap, const float* __restrict__ flow,
                                int flow_cs, float* __restrict__ out, int out_cs, int N, int H, int W, int C, float stepx,
                                float stepy, float sclx, float scly) {
    const int CQ = VEC ? C / 4 : 1;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * H * W * CQ) return;
    const int q = idx % CQ;
    const long p = idx / CQ;
    const int x = p % W, y = (p / W) % H;
    const int n = p / ((long)W * H);
    const WarpTap t = m2m_taps(x, y, flow[p * flow_cs], flow[p * flow_cs + 1], H, W, stepx, stepy, sclx, scly);
    const float* b = in + (size_t)(in_swap ? (n ^ 1) : n) * H * W * in_cs;
    float* o = out + p * out_cs;
    if (VEC) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t.off[k] >= 0) {
                const float4 v = *(const float4*)(b + (size_t)t.off[k] * in_cs + q * 4);
                r.x = __fadd_rn(r.x, __fmul_rn(v.x, t.w[k]));
                r.y = __fadd_rn(r.y, __fmul_rn(v.y, t.w[k]));
                r.z = __fadd_rn(r.z, __fmul_rn(v.z, t.w[k]));
                r.w = __fadd_rn(r.w, __fmul_rn(v.w, t.w[k]));
            }
        *(float4*)(o + q * 4) = r;
    } else {
        for (int c = 0; c < C; ++c) o[c] = tap_acc(t, b, in_cs, c);
    }
}

// ---- pooled means for the cube (adaptive_avg_pool2d to Hx1, 1xW, 1x1) ----------------------------------------------
// One block per output row of the pooled tensor, one thread per channel (coalesced over channels), double accumulation.
// mode 1: out[n, y, c] = mean_x in[n,y,x,c];  mode 2: out[n, x, c] = mean_y in[n,y,x,c];  mode 0: out[n, 0, c] = mean_yx.
__global__ void pool_mean_kernel(const float* __restrict__ in, int in_cs, float* __restrict__ out, int out_cs, int H, int W,
                                 int C, int mode) {
    const int L = mode == 1 ? H : W;
    const int n = blockIdx.x / L, l = blockIdx.x - n * L;
    const float* b = in + (size_t)n * H * W * in_cs;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s = 0.0;
        if (mode == 1) {
            for (int x = 0; x < W; ++x) s += b[((size_t)l * W + x) * in_cs + c];
            s /= W;
        } else {
            for (int y = 0; y < H; ++y) s += b[((size_t)y * W + l) * in_cs + c];
            s /= H;
        }
        out[((size_t)n * L + l) * out_cs + c] = (float)s;
    }
}
// global pool: one block per (n, 64-channel group); 4 pixel strips per block reduced through LDS
__global__ __launch_bounds__(256) void pool_global_kernel(const float* __restrict__ in, int in_cs, float* __restrict__ out,
                                                          int out_cs, int H, int W, int C) {
    __shared__ double red[4][64];
    const int n = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), strip = threadIdx.x >> 6;
    const float* b = in + (size_t)n * H * W * in_cs;
    double s = 0.0;
    if (c < C)
        for (long p = strip; p < (long)H * W; p += 4) s += b[p * in_cs + c];
    red[strip][threadIdx.x & 63] = s;
    __syncthreads();
    if (strip == 0 && c < C)
        out[(size_t)n * out_cs + c] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) /
                                              ((double)H * W));
}

// mode 1 / 2 with FEW channels (IFRNet's per-image RGB mean at 1080p: row means of a [2, 1088, 1920, 4] tensor, then the mean of the
// rows; ifrnet/IFRNet_L_arch.py: mean_ = cat(img0, img1).mean): pool_mean_kernel gives every row C = 4 working threads that walk
// 1920 pixels each (0.37 ms, 13 % of an IFRNet_S frame).  Here the 256 threads of a row's block are 256 / C pixel lanes x C channels.
__global__ __launch_bounds__(256) void pool_mean_fewc_kernel(const float* __restrict__ in, int in_cs, float* __restrict__ out, int out_cs,
                                                             int H, int W, int C, int mode) {
    __shared__ double red[256];
    const int L = mode == 1 ? H : W, M = mode == 1 ? W : H;      // L output rows, each the mean over M pixels
    const int n = blockIdx.x / L, l = blockIdx.x - n * L, tid = threadIdx.x;
    const int P = 256 / C, p = tid / C, c = tid - p * C;
    const float* b = in + (size_t)n * H * W * in_cs + c;
    const size_t first = mode == 1 ? (size_t)l * W : (size_t)l, step = mode == 1 ? 1 : (size_t)W;
    double s = 0.0;
    if (p < P)
        for (int i = p; i < M; i += P) s += b[(first + i * step) * in_cs];
    red[tid] = s;
    __syncthreads();
    if (tid < C) {
        for (int k = 1; k < P; ++k) s += red[tid + k * C];
        out[((size_t)n * L + l) * out_cs + tid] = (float)(s / M);
    }
}

// out[n,y,x,c] = s3[n,y,x,c] * mean_k( cC[n, k*C + c] * cH[n, y, k] * cW[n, x, k] ),  k < 16   (M2M_arch.py:786-795)
__global__ void cube_apply_kernel(const float* __restrict__ s3, int s_cs, const float* __restrict__ cC, const float* __restrict__ cH,
                                  int h_cs, const float* __restrict__ cW, int w_cs, float* __restrict__ out, int out_cs, int N,
                                  int H, int W, int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * H * W * C) return;
    const int c = idx % C;
    const long p = idx / C;
    const int x = p % W, y = (p / W) % H;
    const int n = p / ((long)W * H);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float v = __fmul_rn(__fmul_rn(cC[(size_t)n * 16 * C + k * C + c], cH[((size_t)n * H + y) * h_cs + k]),
                                  cW[((size_t)n * W + x) * w_cs + k]);
        acc = __fadd_rn(acc, v);
    }
    out[p * out_cs + c] = __fmul_rn(s3[p * s_cs + c], __fdiv_rn(acc, 16.0f));
}

// ---- splat preparation ------------------------------------------------------------------------------------------
// Per pair (timestep independent), for direction d (image n of the batch of 2) and branch b < 4, splat s = 2*b + d:
//   tf_s    = flow_d + res_d[2b:2b+2]                                           (M2M_arch.py:945-958)
//   photo_s = clip(1 - wei_d * mean_c |im_d - backwarp(im_other, tf_s)|, 0.001)^2    (:987-1010), wei = sigmoid*0.8+0.1 (:842-846)
//   E_s     = exp(clip(alpha * photo_s, -20, 20))                                (:559-561)
// d0: [2,H,W,d0_cs] = (flow 0..1 | image 2..4 | ..);  r: [2,H,W,r_cs] = (8 flow residuals | mask logit)
__global__ void m2m_photo_kernel(const float* __restrict__ d0, int d0_cs, const float* __restrict__ r, int r_cs, float alpha,
                                 float* __restrict__ TF, float* __restrict__ E, int H, int W, float stepx, float stepy, float sclx,
                                 float scly) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long hw = (long)H * W;
    if (idx >= 2 * hw) return;
    const int d = idx / hw;
    const long p = idx - d * hw;
    const int x = p % W, y = p / W;
    const float* me = d0 + (size_t)idx * d0_cs;
    const float* other = d0 + (size_t)(d ^ 1) * hw * d0_cs + 2;
    const float* rr = r + (size_t)idx * r_cs;
    const float wei = __fadd_rn(__fmul_rn(1.0f / (1.0f + expf(-rr[8])), 0.8f), 0.1f);
    const float i0 = me[2], i1 = me[3], i2 = me[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float fx = __fadd_rn(me[0], rr[2 * b]), fy = __fadd_rn(me[1], rr[2 * b + 1]);
        const WarpTap t = m2m_taps(x, y, fx, fy, H, W, stepx, stepy, sclx, scly);
        const float a0 = fabsf(__fsub_rn(i0, tap_acc(t, other, d0_cs, 0)));
        const float a1 = fabsf(__fsub_rn(i1, tap_acc(t, other, d0_cs, 1)));
        const float a2 = fabsf(__fsub_rn(i2, tap_acc(t, other, d0_cs, 2)));
        const float m = __fdiv_rn(__fadd_rn(__fadd_rn(a0, a1), a2), 3.0f);
        float ph = fmaxf(__fsub_rn(1.0f, __fmul_rn(wei, m)), 0.001f);
        ph = __fmul_rn(ph, ph);
        const float met = fminf(fmaxf(__fmul_rn(alpha, ph), -20.0f), 20.0f);
        const size_t s = (size_t)(2 * b + d) * hw + p;
        TF[s * 2] = fx;
        TF[s * 2 + 1] = fy;
        E[s] = expf(met);
    }
}
// Per timestep: IN_s = (im_d * td * E_s, td * E_s), FL_s = tf_s * tm   with (td, tm) = (1-t, t) for d = 0, (t, 1-t) for d = 1
__global__ void m2m_splat_inputs_kernel(const float* __restrict__ d0, int d0_cs, const float* __restrict__ TF,
                                        const float* __restrict__ E, float t, float* __restrict__ IN, float* __restrict__ FL,
                                        long hw) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 8 * hw) return;
    const int s = idx / hw;
    const long p = idx - s * hw;
    const int d = s & 1;
    const float t1 = __fsub_rn(1.0f, t);
    const float td = d ? t : t1, tm = d ? t1 : t;
    const float* im = d0 + ((size_t)d * hw + p) * d0_cs + 2;
    const float e = E[idx];
    float4 v;
    v.x = __fmul_rn(__fmul_rn(im[0], td), e);
    v.y = __fmul_rn(__fmul_rn(im[1], td), e);
    v.z = __fmul_rn(__fmul_rn(im[2], td), e);
    v.w = __fmul_rn(td, e);
    *(float4*)(IN + idx * 4) = v;
    *(float2*)(FL + idx * 2) = make_float2(__fmul_rn(TF[idx * 2], tm), __fmul_rn(TF[idx * 2 + 1], tm));
}
// forwarp_mframe_mask accumulation order (:569-581), hole fill (:1026-1031), de-normalisation and crop (:1033-1037)
__global__ void m2m_combine_kernel(const float* __restrict__ O, const float* __restrict__ d0, int d0_cs,
                                   const float* __restrict__ stats, float t, float* __restrict__ out, int Hp, int Wp, int H,
                                   int W) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)H * W) return;
    const int x = idx % W, y = idx / W;
    const long hw = (long)Hp * Wp, p = (long)y * Wp + x;
    float acc[3] = {0.f, 0.f, 0.f}, norm = 0.f;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float4 f = *(const float4*)(O + ((size_t)(2 * b) * hw + p) * 4);
        const float4 g = *(const float4*)(O + ((size_t)(2 * b + 1) * hw + p) * 4);
        acc[0] = __fadd_rn(acc[0], __fadd_rn(f.x, g.x));
        acc[1] = __fadd_rn(acc[1], __fadd_rn(f.y, g.y));
        acc[2] = __fadd_rn(acc[2], __fadd_rn(f.z, g.z));
        norm = __fadd_rn(norm, __fadd_rn(__fadd_rn(f.w, 0.0000001f), __fadd_rn(g.w, 0.0000001f)));
    }
    const float t1 = __fsub_rn(1.0f, t);
    const float* a = d0 + (size_t)p * d0_cs + 2;
    const float* bb = d0 + (size_t)(hw + p) * d0_cs + 2;
    const bool hole = norm < 0.00001f;
    const float mean = stats[0], sd = stats[1];
    float* o = out + (size_t)idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = __fdiv_rn(acc[c], norm);
        if (hole) v = __fadd_rn(v, __fadd_rn(__fmul_rn(t1, a[c]), __fmul_rn(t, bb[c])));
        o[c] = __fadd_rn(__fmul_rn(v, sd), mean);
    }
}

}  // namespace vfi

using namespace vfi;

static void warp_consts(int H, int W, float& stepx, float& stepy, float& sclx, float& scly) { m2m_warp_consts(H, W, stepx, stepy, sclx, scly); }

extern "C" {

int vfi_m2m_normalize(const float* frame0_dev, const float* frame1_dev, int C, int H, int W, int Hp, int Wp, float* out_dev,
                      int out_cs, int out_coff, float* stats_dev, void* workspace_dev, int64_t workspace_bytes, void* stream) {
    VFI_REQUIRE(frame0_dev && frame1_dev && out_dev && stats_dev && workspace_dev && C >= 3 && H > 0 && W > 0 && Hp >= H && Wp >= W,
                "vfi_m2m_normalize: bad arguments");
    VFI_REQUIRE(workspace_bytes >= (int64_t)(ST_BLOCKS * 4 * sizeof(double)), "vfi_m2m_normalize: workspace of %d bytes needed",
                (int)(ST_BLOCKS * 4 * sizeof(double)));
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("m2m_normalize", s);
    hipLaunchKernelGGL(m2m_stats_partial, dim3(ST_BLOCKS), dim3(256), 0, s, frame0_dev, frame1_dev, C, H, W, Hp, Wp,
                       (double*)workspace_dev);
    hipLaunchKernelGGL(m2m_stats_final, dim3(1), dim3(256), 0, s, (const double*)workspace_dev, ST_BLOCKS, (double)Hp * Wp * 3.0,
                       stats_dev);
    hipLaunchKernelGGL(m2m_normalize_kernel, dim3(nblk(2L * Hp * Wp)), dim3(256), 0, s, frame0_dev, frame1_dev, C, H, W, Hp, Wp,
                       stats_dev, out_dev, out_cs, out_coff);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_warp_m2m(const float* in_dev, int in_cs, int in_swap, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N,
                 int H, int W, int C, void* stream) {
    VFI_REQUIRE(in_dev && flow_dev && out_dev && N > 0 && H > 1 && W > 1 && C > 0 && (!in_swap || N % 2 == 0),
                "vfi_warp_m2m: bad arguments");
    float stepx, stepy, sclx, scly;
    warp_consts(H, W, stepx, stepy, sclx, scly);
    const bool vec = C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0 && ((uintptr_t)in_dev & 15) == 0 && ((uintptr_t)out_dev & 15) == 0;
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("warp_m2m", s);
    if (vec)
        hipLaunchKernelGGL(warp_m2m_kernel<true>, dim3(nblk((long)N * H * W * (C / 4))), dim3(256), 0, s, in_dev, in_cs, in_swap,
                           flow_dev, flow_cs, out_dev, out_cs, N, H, W, C, stepx, stepy, sclx, scly);
    else
        hipLaunchKernelGGL(warp_m2m_kernel<false>, dim3(nblk((long)N * H * W)), dim3(256), 0, s, in_dev, in_cs, in_swap, flow_dev,
                           flow_cs, out_dev, out_cs, N, H, W, C, stepx, stepy, sclx, scly);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_pool_mean(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int H, int W, int C, int mode, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && N > 0 && H > 0 && W > 0 && C > 0 && mode >= 0 && mode <= 2, "vfi_pool_mean: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("pool_mean", s);
    if (mode != 0 && C <= 32 && (mode == 1 ? W : H) >= 256) {
        hipLaunchKernelGGL(pool_mean_fewc_kernel, dim3(N * (mode == 1 ? H : W)), dim3(256), 0, s, in_dev, in_cs, out_dev, out_cs, H, W, C, mode);
    } else if (mode == 0) {
        hipLaunchKernelGGL(pool_global_kernel, dim3((C + 63) / 64, N), dim3(256), 0, s, in_dev, in_cs, out_dev, out_cs, H, W, C);
    } else {
        hipLaunchKernelGGL(pool_mean_kernel, dim3(N * (mode == 1 ? H : W)), dim3(C < 256 ? 64 * ((C + 63) / 64) : 256), 0, s, in_dev,
                           in_cs, out_dev, out_cs, H, W, C, mode);
    }
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_m2m_cube_apply(const float* s3_dev, int s3_cs, const float* cC_dev, const float* cH_dev, int cH_cs, const float* cW_dev,
                       int cW_cs, float* out_dev, int out_cs, int N, int H, int W, int C, void* stream) {
    VFI_REQUIRE(s3_dev && cC_dev && cH_dev && cW_dev && out_dev && N > 0 && C > 0, "vfi_m2m_cube_apply: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("cube_apply", s);
    hipLaunchKernelGGL(cube_apply_kernel, dim3(nblk((long)N * H * W * C)), dim3(256), 0, s, s3_dev, s3_cs, cC_dev, cH_dev, cH_cs,
                       cW_dev, cW_cs, out_dev, out_cs, N, H, W, C);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_m2m_photo(const float* d0_dev, int d0_cs, const float* r_dev, int r_cs, float alpha, float* tf_dev, float* e_dev, int H,
                  int W, void* stream) {
    VFI_REQUIRE(d0_dev && r_dev && tf_dev && e_dev && d0_cs >= 5 && r_cs >= 9 && H > 1 && W > 1, "vfi_m2m_photo: bad arguments");
    float stepx, stepy, sclx, scly;
    warp_consts(H, W, stepx, stepy, sclx, scly);
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("m2m_photo", s);
    hipLaunchKernelGGL(m2m_photo_kernel, dim3(nblk(2L * H * W)), dim3(256), 0, s, d0_dev, d0_cs, r_dev, r_cs, alpha, tf_dev, e_dev, H,
                       W, stepx, stepy, sclx, scly);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_m2m_splat_inputs(const float* d0_dev, int d0_cs, const float* tf_dev, const float* e_dev, float t, float* in_dev,
                         float* flow_dev, int H, int W, void* stream) {
    VFI_REQUIRE(d0_dev && tf_dev && e_dev && in_dev && flow_dev && H > 0 && W > 0, "vfi_m2m_splat_inputs: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("m2m_splat_inputs", s);
    hipLaunchKernelGGL(m2m_splat_inputs_kernel, dim3(nblk(8L * H * W)), dim3(256), 0, s, d0_dev, d0_cs, tf_dev, e_dev, t, in_dev,
                       flow_dev, (long)H * W);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_m2m_combine(const float* splat_dev, const float* d0_dev, int d0_cs, const float* stats_dev, float t, float* out_dev, int Hp,
                    int Wp, int H, int W, void* stream) {
    VFI_REQUIRE(splat_dev && d0_dev && stats_dev && out_dev && Hp >= H && Wp >= W && H > 0 && W > 0, "vfi_m2m_combine: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("m2m_combine", s);
    hipLaunchKernelGGL(m2m_combine_kernel, dim3(nblk((long)H * W)), dim3(256), 0, s, splat_dev, d0_dev, d0_cs, stats_dev, t, out_dev,
                       Hp, Wp, H, W);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
