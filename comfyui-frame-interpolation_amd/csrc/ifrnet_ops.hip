// Kernels of the IFRNet path (vfi_models/ifrnet/IFRNet_L_arch.py, IFRNet_S_arch.py) that are not generic layer calls:
// input assembly and mean removal (forward :230-249), the L model's 7x7 stride-2 head conv (Encoder :128-130), the
// time-embedding plane (Decoder4 :160-163), sigmoid on a channel window (:278), F.interpolate with an explicit
// scale_factor (:38-41,281-288) and the output kernel: both image warps, mask blend, mean, residual, clamp, crop (:290-294).
// ifrnet.py drives them together with vfi_conv_forward_ex / vfi_warp_rife / vfi_axpby / vfi_pool_mean.
#include <cmath>

#include "../../include/vfi_hip.h"
#include "rife_warp.h"

namespace vfi {

static unsigned nblk_i(long n) { return (unsigned)((n + 255) / 256); }

// out{0,1}[y,x] = (frame{0,1}.rgb, 0) inside H x W, zeros in the padding (F.pad, :230-235; no clamp on this path)
__global__ void ifrnet_prep_kernel(const float* __restrict__ f0, const float* __restrict__ f1, int C, int H, int W,
                                   float* __restrict__ out0, float* __restrict__ out1, int Hp, int Wp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hp * Wp) return;
    const int x = idx % Wp, y = idx / Wp;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (y < H && x < W) {
        const float* pa = f0 + ((size_t)y * W + x) * C;
        const float* pb = f1 + ((size_t)y * W + x) * C;
        a = make_float4(pa[0], pa[1], pa[2], 0.f);
        b = make_float4(pb[0], pb[1], pb[2], 0.f);
    }
    ((float4*)out0)[idx] = a;
    ((float4*)out1)[idx] = b;
}

// mean_[n] = mean over both padded images and their 3 channels (:242-247), from the per-image channel means
// cm [2N][4] (vfi_pool_mean mode 0 over img [2N,Hp,Wp,4], images 0..N-1 = img0, N..2N-1 = img1);
// img{0,1}[n] -= mean_[n] on the 3 colour channels (the zero padding becomes -mean, like the reference's).
__global__ void ifrnet_center_kernel(float* __restrict__ img, const float* __restrict__ cm, float* __restrict__ mean_out, int N,
                                     long px) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = blockIdx.y;           // image index 0..2N-1
    const int n = k % N;
    const float* a = cm + (size_t)n * 4;
    const float* b = cm + (size_t)(N + n) * 4;
    const float m = (((a[0] + a[1]) + a[2]) + ((b[0] + b[1]) + b[2])) / 6.0f;
    if (idx == 0 && k < N) mean_out[n] = m;
    if (idx >= px) return;
    float4* p = (float4*)img + (size_t)k * px + idx;
    float4 v = *p;
    v.x -= m;
    v.y -= m;
    v.z -= m;
    *p = v;
}

// Conv2d(3, CO, 7, 2, 3) + PReLU(CO): one thread per output pixel, all CO channels in registers, weights
// [ky][kx][ci][co] read through the scalar cache (uniform addresses).  3 input channels cannot feed the MFMA path
// (K = 147 per pixel); 0.3 GFLOP per 1080p half-resolution image pair, so VALU is fine here.
template <int CO>
__global__ __launch_bounds__(128) void conv7x7s2_prelu_kernel(const float* __restrict__ in, int in_cs, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ slope,
                                                             float* __restrict__ out, int out_cs, int N, int Hin, int Win) {
    const int Ho = (Hin - 1) / 2 + 1, Wo = (Win - 1) / 2 + 1;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * Ho * Wo) return;
    const int ox = idx % Wo, oy = (idx / Wo) % Ho;
    const int n = idx / ((long)Wo * Ho);
    float acc[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) acc[co] = bias[co];
    const float* b = in + (size_t)n * Hin * Win * in_cs;
    for (int ky = 0; ky < 7; ++ky) {
        const int iy = 2 * oy - 3 + ky;
        for (int kx = 0; kx < 7; ++kx) {
            const int ix = 2 * ox - 3 + kx;
            const bool ok = iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            const float* p = b + ((size_t)(ok ? iy : 0) * Win + (ok ? ix : 0)) * in_cs;
            const float v0 = ok ? p[0] : 0.f, v1 = ok ? p[1] : 0.f, v2 = ok ? p[2] : 0.f;
            const float* wt = w + (size_t)(ky * 7 + kx) * 3 * CO;
#pragma unroll
            for (int co = 0; co < CO; ++co) acc[co] = fmaf(v2, wt[2 * CO + co], fmaf(v1, wt[CO + co], fmaf(v0, wt[co], acc[co])));
        }
    }
    float* o = out + (size_t)idx * out_cs;
#pragma unroll
    for (int q = 0; q < CO / 4; ++q) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = acc[4 * q + j];
            v[j] = a > 0.f ? a : a * slope[4 * q + j];
        }
        *(float4*)(o + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// x[p, 0..C) = sigmoid(x[p, 0..C)) over a channel window (torch.sigmoid, :278)
__global__ void sigmoid_kernel(float* __restrict__ x, int cs, int C, long px) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= px * C) return;
    const long p = idx / C;
    float* q = x + p * cs + (idx - p * C);
    *q = 1.0f / (1.0f + expf(-*q));
}

// out[n, p, 0..C) = value[n]: the time-embedding plane embt.repeat(1, 1, h, w) (:160-163)
struct FillVals {
    float v[64];
};
__global__ void fill_items_kernel(float* __restrict__ out, int cs, int C, long px_per_item, FillVals vals) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= px_per_item * C) return;
    const long p = idx / C;
    out[((size_t)blockIdx.y * px_per_item + p) * cs + (idx - p * C)] = vals.v[blockIdx.y];
}

// torch area_pixel_compute_source_index(ratio, dst, align_corners=False) + guard_index_and_lambda, with the ratio the
// caller derived from the user's scale_factor (F.interpolate(scale_factor=s) passes 1/s, not in/out)
struct BilS {
    int i0, i1;
    float w0, w1;
};
__device__ static inline BilS bil_src(int d, float ratio, int in_size) {
    float src = __fsub_rn(__fmul_rn(ratio, __fadd_rn((float)d, 0.5f)), 0.5f);
    if (src < 0.f) src = 0.f;
    int i0 = (int)floorf(src);
    if (i0 > in_size - 1) i0 = in_size - 1;
    const float l = fminf(fmaxf(__fsub_rn(src, (float)i0), 0.f), 1.f);
    BilS b;
    b.i0 = i0;
    b.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    b.w1 = l;
    b.w0 = __fsub_rn(1.0f, l);
    return b;
}
__global__ void resize_ratio_kernel(const float* __restrict__ in, int in_cs, float* __restrict__ out, int out_cs, int N, int Hi,
                                    int Wi, int Ho, int Wo, int C, float ry, float rx, float post_mul) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * Ho * Wo) return;
    const int x = idx % Wo, y = (idx / Wo) % Ho;
    const int n = idx / ((long)Wo * Ho);
    const BilS by = bil_src(y, ry, Hi), bx = bil_src(x, rx, Wi);
    const float* b = in + (size_t)n * Hi * Wi * in_cs;
    float* o = out + (size_t)idx * out_cs;
    for (int c = 0; c < C; ++c) {
        const float a = b[((size_t)by.i0 * Wi + bx.i0) * in_cs + c], bb = b[((size_t)by.i0 * Wi + bx.i1) * in_cs + c];
        const float cc = b[((size_t)by.i1 * Wi + bx.i0) * in_cs + c], d = b[((size_t)by.i1 * Wi + bx.i1) * in_cs + c];
        const float v = __fadd_rn(__fmul_rn(by.w0, __fadd_rn(__fmul_rn(bx.w0, a), __fmul_rn(bx.w1, bb))),
                                  __fmul_rn(by.w1, __fadd_rn(__fmul_rn(bx.w0, cc), __fmul_rn(bx.w1, d))));
        o[c] = __fmul_rn(v, post_mul);
    }
}

// imgt = clamp(mask * warp(img0, flow0) + (1 - mask) * warp(img1, flow1) + mean_ + res, 0, 1)[:H, :W]   (:290-294)
// fin [N,Hf,Wf,8] = (flow0 xy, flow1 xy, mask (already sigmoid), res rgb) at the resolution the reference's last resize
// produced; the warp grid has THAT size while the images keep Hp x Wp (warp() builds its grid from the flow, :10-13,
// and grid_sample un-normalises with the image size).
__global__ void ifrnet_output_kernel(const float* __restrict__ img0, const float* __restrict__ img1, const float* __restrict__ fin,
                                     const float* __restrict__ mean, float* __restrict__ out, int Hp, int Wp, int Hf, int Wf, int H,
                                     int W) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int n = blockIdx.y;
    const int X = idx % W, Y = idx / W;
    const WarpGeo gf = make_warp_geo(Wf, Hf), gi = make_warp_geo(Wp, Hp);
    const float* f = fin + ((size_t)n * Hf * Wf + (size_t)Y * Wf + X) * 8;
    const float4 fl = *(const float4*)f, mr = *(const float4*)(f + 4);
    const float m = mr.x, om = __fsub_rn(1.0f, m), mean_n = mean[n];
    const float resv[3] = {mr.y, mr.z, mr.w};
    float wv[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float fx = k ? fl.z : fl.x, fy = k ? fl.w : fl.y;
        // grid = linspace(-1,1,Wf)[X] + fx/((Wf-1)/2); ix = ((grid+1)/2)*(Wp-1), border clamp, align_corners=True
        const float nx = __fadd_rn(lin11(X, gf.W, gf.stepx), __fdiv_rn(fx, gf.halfw));
        const float ny = __fadd_rn(lin11(Y, gf.H, gf.stepy), __fdiv_rn(fy, gf.halfh));
        float px = __fmul_rn(__fadd_rn(nx, 1.0f), gi.halfw);
        float py = __fmul_rn(__fadd_rn(ny, 1.0f), gi.halfh);
        px = fminf((float)(Wp - 1), fmaxf(px, 0.0f));
        py = fminf((float)(Hp - 1), fmaxf(py, 0.0f));
        const float x0f = floorf(px), y0f = floorf(py);
        const float tw = __fsub_rn(px, x0f), te = __fsub_rn(1.0f, tw);
        const float tn = __fsub_rn(py, y0f), ts = __fsub_rn(1.0f, tn);
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int x1 = x0 + (x0 < Wp - 1 ? 1 : 0), y1 = y0 + (y0 < Hp - 1 ? 1 : 0);
        const float* im = (k ? img1 : img0) + (size_t)n * Hp * Wp * 4;
        const float4 a = *(const float4*)(im + ((size_t)y0 * Wp + x0) * 4), b = *(const float4*)(im + ((size_t)y0 * Wp + x1) * 4);
        const float4 c = *(const float4*)(im + ((size_t)y1 * Wp + x0) * 4), d = *(const float4*)(im + ((size_t)y1 * Wp + x1) * 4);
        const float nw = __fmul_rn(ts, te), ne = __fmul_rn(ts, tw), sw = __fmul_rn(tn, te), se = __fmul_rn(tn, tw);
        wv[k][0] = a.x * nw + b.x * ne + c.x * sw + d.x * se;
        wv[k][1] = a.y * nw + b.y * ne + c.y * sw + d.y * se;
        wv[k][2] = a.z * nw + b.z * ne + c.z * sw + d.z * se;
    }
    float* o = out + ((size_t)n * H * W + idx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float merged = __fadd_rn(__fadd_rn(__fmul_rn(m, wv[0][c]), __fmul_rn(om, wv[1][c])), mean_n);
        o[c] = fminf(fmaxf(__fadd_rn(merged, resv[c]), 0.f), 1.f);
    }
}

}  // namespace vfi

using namespace vfi;

extern "C" {

int vfi_ifrnet_prep(const float* frame0_dev, const float* frame1_dev, int C, int H, int W, float* img0_dev, float* img1_dev, int Hp,
                    int Wp, void* stream) {
    VFI_REQUIRE(frame0_dev && frame1_dev && img0_dev && img1_dev && C >= 3 && H > 0 && W > 0 && Hp >= H && Wp >= W,
                "vfi_ifrnet_prep: bad arguments");
    TraceScope ts("ifrnet_prep", (hipStream_t)stream);
    hipLaunchKernelGGL(ifrnet_prep_kernel, dim3(nblk_i((long)Hp * Wp)), dim3(256), 0, (hipStream_t)stream, frame0_dev, frame1_dev, C, H,
                       W, img0_dev, img1_dev, Hp, Wp);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_ifrnet_center(float* img_dev, const float* chan_means_dev, float* mean_out_dev, int N, int64_t pixels, void* stream) {
    VFI_REQUIRE(img_dev && chan_means_dev && mean_out_dev && N > 0 && pixels > 0, "vfi_ifrnet_center: bad arguments");
    TraceScope ts("ifrnet_center", (hipStream_t)stream);
    hipLaunchKernelGGL(ifrnet_center_kernel, dim3(nblk_i(pixels), 2 * N), dim3(256), 0, (hipStream_t)stream, img_dev, chan_means_dev,
                       mean_out_dev, N, (long)pixels);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_conv7x7s2_prelu(const float* in_dev, int in_cs, const float* w_dev, const float* bias_dev, const float* slope_dev, int Cout,
                        float* out_dev, int out_cs, int N, int Hin, int Win, void* stream) {
    VFI_REQUIRE(in_dev && w_dev && bias_dev && slope_dev && out_dev && N > 0 && Hin > 0 && Win > 0 && in_cs >= 3 && out_cs >= Cout &&
                    out_cs % 4 == 0 && ((uintptr_t)out_dev & 15) == 0,
                "vfi_conv7x7s2_prelu: bad arguments");
    VFI_REQUIRE(Cout == 64, "vfi_conv7x7s2_prelu: Cout=%d not instantiated (IFRNet_L's head has 64)", Cout);
    const int Ho = (Hin - 1) / 2 + 1, Wo = (Win - 1) / 2 + 1;
    TraceScope ts("conv7x7s2", (hipStream_t)stream);
    hipLaunchKernelGGL((conv7x7s2_prelu_kernel<64>), dim3((unsigned)(((long)N * Ho * Wo + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                       in_dev, in_cs, w_dev, bias_dev, slope_dev, out_dev, out_cs, N, Hin, Win);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_sigmoid(float* x_dev, int cs, int C, int64_t pixels, void* stream) {
    VFI_REQUIRE(x_dev && C > 0 && cs >= C && pixels > 0, "vfi_sigmoid: bad arguments");
    TraceScope ts("sigmoid", (hipStream_t)stream);
    hipLaunchKernelGGL(sigmoid_kernel, dim3(nblk_i(pixels * C)), dim3(256), 0, (hipStream_t)stream, x_dev, cs, C, (long)pixels);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_fill_items(float* out_dev, int cs, int C, int N, int64_t pixels_per_item, const float* values_host, void* stream) {
    VFI_REQUIRE(out_dev && values_host && C > 0 && cs >= C && N > 0 && N <= 64 && pixels_per_item > 0, "vfi_fill_items: bad arguments");
    FillVals fv;
    for (int i = 0; i < 64; ++i) fv.v[i] = i < N ? values_host[i] : 0.f;
    TraceScope ts("fill_items", (hipStream_t)stream);
    hipLaunchKernelGGL(fill_items_kernel, dim3(nblk_i(pixels_per_item * C), N), dim3(256), 0, (hipStream_t)stream, out_dev, cs, C,
                       (long)pixels_per_item, fv);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_resize_bilinear_ratio(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int Hin, int Win, int Hout, int Wout,
                              int C, float ratio_h, float ratio_w, float post_mul, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && C > 0 && in_cs >= C && out_cs >= C && N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 &&
                    ratio_h > 0.f && ratio_w > 0.f,
                "vfi_resize_bilinear_ratio: bad arguments");
    TraceScope ts("resize_ratio", (hipStream_t)stream);
    hipLaunchKernelGGL(resize_ratio_kernel, dim3(nblk_i((long)N * Hout * Wout)), dim3(256), 0, (hipStream_t)stream, in_dev, in_cs, out_dev,
                       out_cs, N, Hin, Win, Hout, Wout, C, ratio_h, ratio_w, post_mul);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_ifrnet_output(const float* img0_dev, const float* img1_dev, const float* fin_dev, const float* mean_dev, float* out_dev, int N,
                      int Hp, int Wp, int Hf, int Wf, int H, int W, void* stream) {
    VFI_REQUIRE(img0_dev && img1_dev && fin_dev && mean_dev && out_dev && N > 0 && Hp >= H && Wp >= W && Hf >= H && Wf >= W && Hf > 1 &&
                    Wf > 1,
                "vfi_ifrnet_output: bad arguments (the flow field must cover the %dx%d frame: %dx%d)", H, W, Hf, Wf);
    TraceScope ts("ifrnet_output", (hipStream_t)stream);
    hipLaunchKernelGGL(ifrnet_output_kernel, dim3(nblk_i((long)H * W), N), dim3(256), 0, (hipStream_t)stream, img0_dev, img1_dev, fin_dev,
                       mean_dev, out_dev, Hp, Wp, Hf, Wf, H, W);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
