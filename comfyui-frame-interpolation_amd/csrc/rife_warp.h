// Backward-warp geometry shared by the RIFE kernels: the reference's warp() (vfi_models/rife/rife_arch.py:31-70) in its
// own fp32 expression order (normalise -> grid_sample un-normalise round trip, border clamp, align_corners=True).
#pragma once
#include "vfi_common.h"

namespace vfi {

// ---------------------------------------------------------------------------------------
// bilinear backward warp in the reference's fp32 expression order
// ---------------------------------------------------------------------------------------
struct WarpGeo {
    int W, H;
    float stepx, stepy;    // 2/(W-1), 2/(H-1): torch.linspace step
    float halfw, halfh;    // (W-1)/2, (H-1)/2
};
__host__ __device__ static inline WarpGeo make_warp_geo(int W, int H) {
    WarpGeo g;
    g.W = W;
    g.H = H;
    g.stepx = 2.0f / (float)(W - 1);
    g.stepy = 2.0f / (float)(H - 1);
    g.halfw = (float)((W - 1.0) / 2.0);
    g.halfh = (float)((H - 1.0) / 2.0);
    return g;
}

struct Tap4 {
    int o00, o01, o10, o11;   // pixel indices (y*W+x) of the 4 taps
    float nw, ne, sw, se;
};

// torch.linspace(-1,1,n)[i]: start + step*i below the midpoint, end - step*(n-1-i) above it.
__device__ static inline float lin11(int i, int n, float step) {
    return i < n / 2 ? __fadd_rn(-1.0f, __fmul_rn(step, (float)i))
                     : __fsub_rn(1.0f, __fmul_rn(step, (float)(n - 1 - i)));
}

// grid = base + flow/((size-1)/2); grid_sample(align_corners=True, border):
//   ix = (g+1)*((size-1)/2), clipped to [0,size-1]; corner weights (1-tx)(1-ty) ...
__device__ static inline Tap4 warp_taps(const WarpGeo& g, int X, int Y, float fx, float fy) {
    const float nx = __fadd_rn(lin11(X, g.W, g.stepx), __fdiv_rn(fx, g.halfw));
    const float ny = __fadd_rn(lin11(Y, g.H, g.stepy), __fdiv_rn(fy, g.halfh));
    float px = __fmul_rn(__fadd_rn(nx, 1.0f), g.halfw);
    float py = __fmul_rn(__fadd_rn(ny, 1.0f), g.halfh);
    px = fminf((float)(g.W - 1), fmaxf(px, 0.0f));
    py = fminf((float)(g.H - 1), fmaxf(py, 0.0f));
    const float x0f = floorf(px), y0f = floorf(py);
    const float w = __fsub_rn(px, x0f), e = __fsub_rn(1.0f, w);
    const float n = __fsub_rn(py, y0f), s = __fsub_rn(1.0f, n);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x1 = x0 + (x0 < g.W - 1 ? 1 : 0), y1 = y0 + (y0 < g.H - 1 ? 1 : 0);
    Tap4 t;
    t.o00 = y0 * g.W + x0;
    t.o01 = y0 * g.W + x1;
    t.o10 = y1 * g.W + x0;
    t.o11 = y1 * g.W + x1;
    t.nw = __fmul_rn(s, e);
    t.ne = __fmul_rn(s, w);
    t.sw = __fmul_rn(n, e);
    t.se = __fmul_rn(n, w);
    return t;
}

}  // namespace vfi
