// M2M's backwarp taps (grid_sample bilinear, zeros padding, align_corners=True; vfi_models/m2m/M2M_arch.py:24-92) — shared by
// m2m_net.hip (warp_m2m, the photometric kernel) and m2m_render.hip (the tiled photometric kernel).
#pragma once
#include <hip/hip_runtime.h>

namespace vfi {

// ---- backwarp: bilinear, zeros padding, align_corners=True --------------------------------------------------
__device__ static inline float lin_m2m(int i, int n, float step) {  // torch.linspace(-1, 1, n)[i] in fp32
    return i < n / 2 ? __fadd_rn(-1.0f, __fmul_rn(step, (float)i)) : __fsub_rn(1.0f, __fmul_rn(step, (float)(n - 1 - i)));
}
struct WarpTap {
    int off[4];    // pixel offsets (y*W+x) of nw, ne, sw, se; -1 when outside
    float w[4];
};
__device__ static inline WarpTap m2m_taps(int x, int y, float fx, float fy, int H, int W, float stepx, float stepy, float sclx,
                                          float scly) {
    const float gx = __fadd_rn(lin_m2m(x, W, stepx), __fmul_rn(fx, sclx));
    const float gy = __fadd_rn(lin_m2m(y, H, stepy), __fmul_rn(fy, scly));
    // grid_sampler unnormalize, align_corners=True: (g + 1) * ((size - 1) / 2); weights as torch's CPU kernel forms
    // them: w = ix - floor(ix), e = 1 - w, n = iy - floor(iy), s = 1 - n; nw = s*e, ne = s*w, sw = n*e, se = n*w
    const float ix = __fmul_rn(__fadd_rn(gx, 1.0f), __fdiv_rn((float)(W - 1), 2.0f));
    const float iy = __fmul_rn(__fadd_rn(gy, 1.0f), __fdiv_rn((float)(H - 1), 2.0f));
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float x1f = __fadd_rn(x0f, 1.0f), y1f = __fadd_rn(y0f, 1.0f);
    const float ww = __fsub_rn(ix, x0f), ee = __fsub_rn(1.0f, ww), nn = __fsub_rn(iy, y0f), ss = __fsub_rn(1.0f, nn);
    WarpTap t;
    t.w[0] = __fmul_rn(ss, ee);
    t.w[1] = __fmul_rn(ss, ww);
    t.w[2] = __fmul_rn(nn, ee);
    t.w[3] = __fmul_rn(nn, ww);
    // non-finite coordinates: every comparison below is false -> all taps dropped (torch yields NaN there; the
    // reference never produces them on finite inputs)
    const bool xin0 = x0f >= 0.f && x0f <= (float)(W - 1), xin1 = x1f >= 0.f && x1f <= (float)(W - 1);
    const bool yin0 = y0f >= 0.f && y0f <= (float)(H - 1), yin1 = y1f >= 0.f && y1f <= (float)(H - 1);
    const int x0 = xin0 ? (int)x0f : 0, x1 = xin1 ? (int)x1f : 0, y0 = yin0 ? (int)y0f : 0, y1 = yin1 ? (int)y1f : 0;
    t.off[0] = xin0 && yin0 ? y0 * W + x0 : -1;
    t.off[1] = xin1 && yin0 ? y0 * W + x1 : -1;
    t.off[2] = xin0 && yin1 ? y1 * W + x0 : -1;
    t.off[3] = xin1 && yin1 ? y1 * W + x1 : -1;
    return t;
}
// torch accumulates nw, ne, sw, se in that order starting from 0
__device__ static inline float tap_acc(const WarpTap& t, const float* __restrict__ b, int cs, int c) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (t.off[k] >= 0) r = __fadd_rn(r, __fmul_rn(b[(size_t)t.off[k] * cs + c], t.w[k]));
    return r;
}

// M2M_arch.py:62-81 — torch.linspace steps and the flow -> grid scales
static inline void m2m_warp_consts(int H, int W, float& stepx, float& stepy, float& sclx, float& scly) {
    stepx = 2.0f / (float)(W - 1);  // torch.linspace step: (end - start) / (steps - 1) in fp32
    stepy = 2.0f / (float)(H - 1);
    // M2M_arch.py:62-81: square inputs scale both components by 2/(H-1), otherwise by (2/(W-1), 2/(H-1))
    sclx = (float)(2.0 / ((double)W - 1.0));
    scly = (float)(2.0 / ((double)H - 1.0));
}

}  // namespace vfi
