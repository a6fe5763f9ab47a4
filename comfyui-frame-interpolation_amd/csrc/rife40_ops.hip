// Small kernels of the RIFE arch "4.0" path (sudo_rife4 checkpoint), which rife40.py drives op by op over the generic
// layer objects: input assembly, windowed border warp, |flow| maximum for the scale-doubling test, output blend.
// Reference: vfi_models/rife/rife_arch.py — forward :476-499,598-607,703-732; warp :31-70.
#include <cmath>

#include "../../include/vfi_hip.h"
#include "rife_warp.h"

namespace vfi {

static unsigned nblk40(long n) { return (unsigned)((n + 255) / 256); }

// out[y,x] = (clamp(f0.rgb,0,1), clamp(f1.rgb,0,1), t, 0) inside H x W; (0,0,0, 0,0,0, t, 0) in the zero padding
// (torch.clamp + F.pad + timestep.repeat, rife_arch.py:476-499)
__global__ void rife40_prep_kernel(const float* __restrict__ f0, const float* __restrict__ f1, int C, int H, int W, float t,
                                   float* __restrict__ out, int Hp, int Wp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Hp * Wp) return;
    const int x = idx % Wp, y = idx / Wp;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, t, 0.f};
    if (y < H && x < W) {
        const float* a = f0 + ((size_t)y * W + x) * C;
        const float* b = f1 + ((size_t)y * W + x) * C;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[c] = fminf(fmaxf(a[c], 0.f), 1.f);
            v[3 + c] = fminf(fmaxf(b[c], 0.f), 1.f);
        }
    }
    float4* o = (float4*)(out + (size_t)idx * 8);
    o[0] = make_float4(v[0], v[1], v[2], v[3]);
    o[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// generic NHWC border warp over channel windows
__global__ void warp_rife_kernel(const float* __restrict__ in, int in_cs, const float* __restrict__ flow, int flow_cs,
                                 float* __restrict__ out, int out_cs, int N, int H, int W, int C) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * H * W) return;
    const int X = idx % W, Y = (idx / W) % H;
    const int n = idx / ((long)W * H);
    const WarpGeo g = make_warp_geo(W, H);
    const Tap4 t = warp_taps(g, X, Y, flow[idx * flow_cs], flow[idx * flow_cs + 1]);
    const float* b = in + (size_t)n * H * W * in_cs;
    float* o = out + (size_t)idx * out_cs;
    for (int c = 0; c < C; ++c)
        o[c] = b[(size_t)t.o00 * in_cs + c] * t.nw + b[(size_t)t.o01 * in_cs + c] * t.ne + b[(size_t)t.o10 * in_cs + c] * t.sw +
               b[(size_t)t.o11 * in_cs + c] * t.se;
}

// the same for a compile-time channel count (IFUNet / RIFE 4.0 warp RGB images: C = 3): the channel loop unrolls, the 4 x C loads are
// all in flight before the first use (the run-time loop above issues them one dependent group at a time: 114 us per 1080p RGB warp)
template <int CT>
__global__ void warp_rife_c_kernel(const float* __restrict__ in, int in_cs, const float* __restrict__ flow, int flow_cs,
                                   float* __restrict__ out, int out_cs, int N, int H, int W) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * H * W) return;
    const int X = idx % W, Y = (idx / W) % H;
    const int n = idx / ((long)W * H);
    const WarpGeo g = make_warp_geo(W, H);
    const Tap4 t = warp_taps(g, X, Y, flow[idx * flow_cs], flow[idx * flow_cs + 1]);
    const float* b = in + (size_t)n * H * W * in_cs;
    float* o = out + (size_t)idx * out_cs;
    float v00[CT], v01[CT], v10[CT], v11[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        v00[c] = b[(size_t)t.o00 * in_cs + c];
        v01[c] = b[(size_t)t.o01 * in_cs + c];
        v10[c] = b[(size_t)t.o10 * in_cs + c];
        v11[c] = b[(size_t)t.o11 * in_cs + c];
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) o[c] = v00[c] * t.nw + v01[c] * t.ne + v10[c] * t.sw + v11[c] * t.se;
}

// ... and for channel counts that are multiples of 4 on 16-byte aligned windows: one thread per (pixel, 4 channels), float4 taps
__global__ void warp_rife_v4_kernel(const float* __restrict__ in, int in_cs, const float* __restrict__ flow, int flow_cs,
                                    float* __restrict__ out, int out_cs, int N, int H, int W, int CQ) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)N * H * W * CQ) return;
    const int q = idx % CQ;
    const long p = idx / CQ;
    const int X = p % W, Y = (p / W) % H;
    const int n = p / ((long)W * H);
    const WarpGeo g = make_warp_geo(W, H);
    const Tap4 t = warp_taps(g, X, Y, flow[p * flow_cs], flow[p * flow_cs + 1]);
    const float* b = in + (size_t)n * H * W * in_cs + 4 * q;
    const float4 a00 = *(const float4*)(b + (size_t)t.o00 * in_cs), a01 = *(const float4*)(b + (size_t)t.o01 * in_cs);
    const float4 a10 = *(const float4*)(b + (size_t)t.o10 * in_cs), a11 = *(const float4*)(b + (size_t)t.o11 * in_cs);
    float4 r;
    r.x = a00.x * t.nw + a01.x * t.ne + a10.x * t.sw + a11.x * t.se;
    r.y = a00.y * t.nw + a01.y * t.ne + a10.y * t.sw + a11.y * t.se;
    r.z = a00.z * t.nw + a01.z * t.ne + a10.z * t.sw + a11.z * t.se;
    r.w = a00.w * t.nw + a01.w * t.ne + a10.w * t.sw + a11.w * t.se;
    *(float4*)(out + (size_t)p * out_cs + 4 * q) = r;
}

__global__ void absmax_kernel(const float* __restrict__ x, int cs, int C, long px, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < px * C; i += (long)gridDim.x * blockDim.x) {
        const long p = i / C;
        m = fmaxf(m, fabsf(x[p * cs + (i - p * C)]));   // fmaxf drops NaN like torch.max would not; flows are finite here
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out_bits, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// out[b,y,x,:] = clamp(w0 * sigmoid(m) + w1 * (1 - sigmoid(m)) [ -> clamp(. + (res*2 - 1), 0, 1) ], 0, 1), cropped to H x W
__global__ void rife40_output_kernel(const float* __restrict__ w01, int w_cs, const float* __restrict__ mask, int m_cs,
                                     const float* __restrict__ res, int r_cs, float* __restrict__ out, int Hp, int Wp, int H, int W) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * W) return;
    const int b = blockIdx.y;
    const int x = idx % W, y = idx / W;
    const size_t p = (size_t)b * Hp * Wp + (size_t)y * Wp + x;
    const float m = 1.0f / (1.0f + expf(-mask[p * m_cs]));
    const float om = 1.0f - m;
    float* o = out + ((size_t)b * H * W + idx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = w01[p * w_cs + c] * m + w01[p * w_cs + 3 + c] * om;
        if (res) v = fminf(fmaxf(v + (res[p * r_cs + c] * 2.0f - 1.0f), 0.f), 1.f);
        o[c] = fminf(fmaxf(v, 0.f), 1.f);
    }
}

}  // namespace vfi

using namespace vfi;

extern "C" {

int vfi_rife40_prep(const float* frame0_dev, const float* frame1_dev, int C, int H, int W, float timestep, float* out_dev, int Hp,
                    int Wp, void* stream) {
    VFI_REQUIRE(frame0_dev && frame1_dev && out_dev && C >= 3 && H > 0 && W > 0 && Hp >= H && Wp >= W, "vfi_rife40_prep: bad arguments");
    TraceScope ts("rife40_prep", (hipStream_t)stream);
    hipLaunchKernelGGL(rife40_prep_kernel, dim3(nblk40((long)Hp * Wp)), dim3(256), 0, (hipStream_t)stream, frame0_dev, frame1_dev, C, H,
                       W, timestep, out_dev, Hp, Wp);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_warp_rife(const float* in_dev, int in_cs, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N, int H, int W,
                  int C, void* stream) {
    VFI_REQUIRE(in_dev && flow_dev && out_dev && N > 0 && H > 1 && W > 1 && C > 0 && in_cs >= C && out_cs >= C && flow_cs >= 2,
                "vfi_warp_rife: bad arguments");
    TraceScope ts("warp_rife", (hipStream_t)stream);
    const bool v4 = C % 4 == 0 && in_cs % 4 == 0 && out_cs % 4 == 0 && (((uintptr_t)in_dev | (uintptr_t)out_dev) & 15) == 0;
    if (C == 3)
        hipLaunchKernelGGL(warp_rife_c_kernel<3>, dim3(nblk40((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, in_dev, in_cs, flow_dev, flow_cs,
                           out_dev, out_cs, N, H, W);
    else if (v4)
        hipLaunchKernelGGL(warp_rife_v4_kernel, dim3(nblk40((long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, in_dev, in_cs, flow_dev,
                           flow_cs, out_dev, out_cs, N, H, W, C / 4);
    else
        hipLaunchKernelGGL(warp_rife_kernel, dim3(nblk40((long)N * H * W)), dim3(256), 0, (hipStream_t)stream, in_dev, in_cs, flow_dev,
                           flow_cs, out_dev, out_cs, N, H, W, C);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_absmax(const float* x_dev, int cs, int C, int64_t pixels, float* out_dev, void* stream) {
    VFI_REQUIRE(x_dev && out_dev && C > 0 && cs >= C && pixels > 0, "vfi_absmax: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    VFI_CHECK_HIP(hipMemsetAsync(out_dev, 0, sizeof(float), s));
    TraceScope ts("absmax", s);
    hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, s, x_dev, cs, C, (long)pixels, (unsigned*)out_dev);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int vfi_rife40_output(const float* w01_dev, int w_cs, const float* mask_dev, int m_cs, const float* res_dev, int r_cs, float* out_dev,
                      int B, int Hp, int Wp, int H, int W, void* stream) {
    VFI_REQUIRE(w01_dev && mask_dev && out_dev && B > 0 && Hp >= H && Wp >= W && w_cs >= 6 && m_cs >= 1 && (!res_dev || r_cs >= 3),
                "vfi_rife40_output: bad arguments");
    TraceScope ts("rife40_output", (hipStream_t)stream);
    hipLaunchKernelGGL(rife40_output_kernel, dim3(nblk40((long)H * W), B), dim3(256), 0, (hipStream_t)stream, w01_dev, w_cs, mask_dev, m_cs,
                       res_dev, r_cs, out_dev, Hp, Wp, H, W);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
