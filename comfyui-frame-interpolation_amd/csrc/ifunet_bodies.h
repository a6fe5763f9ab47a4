// Per-element bodies of the IFUNet kernels (second half); same scheme as gmfss_bodies.h: every kernel of
// ifunet_ops.hip runs one of these __host__ __device__ bodies per output element, and tests/hostcheck runs the same bodies on
// the host for the CPU test suite.  Reference: vfi_models/ifunet/IFUNet_arch.py (line numbers below refer to it).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>

#ifndef VFI_HD
#define VFI_HD __host__ __device__ static inline
#endif

namespace vfi_ifunet {

// ---- CBAM channel gate (:411-452): global average and maximum per (n, c), in two passes over `strips` strips ---------------
struct PoolPartArgs {
    const float* x; int cs, C, N; long HW; int strips;
    double* psum; float* pmax;      // [N][strips][C]
};
VFI_HD void chan_pool_partial_body(const PoolPartArgs& a, long idx) {
    if (idx >= (long)a.N * a.strips * a.C) return;
    const int c = (int)(idx % a.C);
    const int s = (int)((idx / a.C) % a.strips);
    const int n = (int)(idx / ((long)a.C * a.strips));
    const long per = (a.HW + a.strips - 1) / a.strips, lo = s * per, hi = lo + per < a.HW ? lo + per : a.HW;
    const float* b = a.x + (size_t)n * a.HW * a.cs + c;
    double sum = 0.0;
    float mx = -INFINITY;
    for (long p = lo; p < hi; ++p) {
        const float v = b[p * a.cs];
        sum += v;
        mx = v > mx ? v : mx;
    }
    a.psum[idx] = sum;
    a.pmax[idx] = mx;
}
struct PoolFinalArgs {
    const double* psum; const float* pmax; int C, N; long HW; int strips;
    float* stats;       // [N][C][2] = mean, max
};
VFI_HD void chan_pool_final_body(const PoolFinalArgs& a, long idx) {
    if (idx >= (long)a.N * a.C) return;
    const int c = (int)(idx % a.C), n = (int)(idx / a.C);
    double sum = 0.0;
    float mx = -INFINITY;
    for (int s = 0; s < a.strips; ++s) {
        const size_t q = ((size_t)n * a.strips + s) * a.C + c;
        sum += a.psum[q];
        mx = a.pmax[q] > mx ? a.pmax[q] : mx;
    }
    a.stats[idx * 2] = (float)(sum / (double)a.HW);
    a.stats[idx * 2 + 1] = mx;
}
// scale[n][c] = sigmoid(mlp(avg)[c] + mlp(max)[c]), mlp = Linear(C, R) -> ReLU -> Linear(R, C)
struct GateArgs {
    const float* stats; const float* w1; const float* b1; const float* w2; const float* b2;   // w1 [R][C], w2 [C][R]
    int C, R, N; float* scale;
};
VFI_HD void cbam_gate_body(const GateArgs& a, long idx) {
    if (idx >= (long)a.N * a.C) return;
    const int c = (int)(idx % a.C), n = (int)(idx / a.C);
    float att = 0.f;
    for (int which = 0; which < 2; ++which) {
        float o = a.b2[c];
        for (int r = 0; r < a.R; ++r) {
            float h = a.b1[r];
            for (int k = 0; k < a.C; ++k) h += a.w1[(size_t)r * a.C + k] * a.stats[((size_t)n * a.C + k) * 2 + which];
            h = h > 0.f ? h : 0.f;
            o += a.w2[(size_t)c * a.R + r] * h;
        }
        att += o;
    }
    a.scale[idx] = 1.0f / (1.0f + expf(-att));
}
// xs = x * scale[n][c]; comp[p] = (max_c xs, mean_c xs): ChannelGate's product and ChannelPool (:451-466)
struct ScaleCompArgs {
    const float* x; int cs; const float* scale; int C, N; long HW;
    float* xs; int xs_cs; float* comp;     // comp [N*HW][2]
};
VFI_HD void cbam_scale_compress_body(const ScaleCompArgs& a, long idx) {
    if (idx >= (long)a.N * a.HW) return;
    const int n = (int)(idx / a.HW);
    const float* b = a.x + idx * a.cs;
    const float* sc = a.scale + (size_t)n * a.C;
    float* o = a.xs + idx * a.xs_cs;
    float mx = -INFINITY, sum = 0.f;
    for (int c = 0; c < a.C; ++c) {
        const float v = b[c] * sc[c];
        o[c] = v;
        mx = v > mx ? v : mx;
        sum += v;
    }
    a.comp[idx * 2] = mx;
    a.comp[idx * 2 + 1] = sum / (float)a.C;
}
// SpatialGate (:469-482): s = BN(conv7x7(comp)) (BatchNorm folded into bn_a, bn_b), out = xs * sigmoid(s); in place on xs
struct SpatialArgs {
    float* xs; int cs; const float* comp; const float* w;   // w [7][7][2]
    float bn_a, bn_b; int C, N, H, W;
};
VFI_HD void cbam_spatial_body(const SpatialArgs& a, long idx) {
    if (idx >= (long)a.N * a.H * a.W) return;
    const int X = (int)(idx % a.W), Y = (int)((idx / a.W) % a.H);
    const int n = (int)(idx / ((long)a.W * a.H));
    const float* cp = a.comp + (size_t)n * a.H * a.W * 2;
    float s = 0.f;
    for (int ky = 0; ky < 7; ++ky) {
        const int y = Y + ky - 3;
        if (y < 0 || y >= a.H) continue;
        for (int kx = 0; kx < 7; ++kx) {
            const int x = X + kx - 3;
            if (x < 0 || x >= a.W) continue;
            const float* q = cp + ((size_t)y * a.W + x) * 2;
            s += q[0] * a.w[(ky * 7 + kx) * 2] + q[1] * a.w[(ky * 7 + kx) * 2 + 1];
        }
    }
    const float g = 1.0f / (1.0f + expf(-(s * a.bn_a + a.bn_b)));
    float* o = a.xs + idx * a.cs;
    for (int c = 0; c < a.C; ++c) o[c] *= g;
}

// ---- IFBlock.upsample_flow (:627-638): convex up-sampling by K of an FC-channel flow ----------------------------------------
struct ConvexUpCArgs {
    const float* mask; int mask_cs;     // [N,H,W, 9*K*K], channel = (j*K + ky)*K + kx
    const float* flow; int flow_cs;     // [N,H,W,FC]
    float* out; int out_cs;             // [N,K*H,K*W,FC]
    int N, H, W, K, FC;
};
VFI_HD void convex_up_c_body(const ConvexUpCArgs& a, long idx) {
    const int KK = a.K * a.K;
    if (idx >= (long)a.N * a.H * a.W * KK) return;
    const int sub = (int)(idx % KK), ky = sub / a.K, kx = sub % a.K;
    const long p = idx / KK;
    const int X = (int)(p % a.W), Y = (int)((p / a.W) % a.H);
    const int n = (int)(p / ((long)a.W * a.H));
    const float* m = a.mask + (size_t)p * a.mask_cs;
    float w[9], mx = -INFINITY;
    for (int j = 0; j < 9; ++j) {
        w[j] = m[(j * a.K + ky) * a.K + kx];
        mx = w[j] > mx ? w[j] : mx;
    }
    float sum = 0.f;
    for (int j = 0; j < 9; ++j) {
        w[j] = expf(w[j] - mx);
        sum += w[j];
    }
    float acc[8];
    for (int c = 0; c < a.FC; ++c) acc[c] = 0.f;
    for (int j = 0; j < 9; ++j) {
        const int x = X + j % 3 - 1, y = Y + j / 3 - 1;       // F.unfold(K * flow, [3,3], padding=1): zero outside
        if (x < 0 || x >= a.W || y < 0 || y >= a.H) continue;
        const float* f = a.flow + (((size_t)n * a.H + y) * a.W + x) * a.flow_cs;
        const float pw = w[j] / sum;
        for (int c = 0; c < a.FC; ++c) acc[c] += pw * ((float)a.K * f[c]);
    }
    float* o = a.out + (((size_t)n * a.H * a.K + (size_t)Y * a.K + ky) * ((size_t)a.W * a.K) + (size_t)X * a.K + kx) * a.out_cs;
    for (int c = 0; c < a.FC; ++c) o[c] = acc[c];
}

// ---- merged = a * mask + b * (1 - mask) (:764) ------------------------------------------------------------------------------
struct LerpArgs {
    const float* a; int a_cs; const float* b; int b_cs; const float* m; int m_cs;
    float* out; int out_cs, C; long px;
};
VFI_HD void lerp_mask_body(const LerpArgs& a, long idx) {
    if (idx >= a.px * a.C) return;
    const long p = idx / a.C;
    const int c = (int)(idx - p * a.C);
    const float m = a.m[p * a.m_cs];
    a.out[p * a.out_cs + c] = a.a[p * a.a_cs + c] * m + a.b[p * a.b_cs + c] * (1.0f - m);
}

// ---- clamp(a + b, 0, 1) (:161) ---------------------------------------------------------------------------------------------
struct AddClampArgs {
    const float* a; int a_cs; const float* b; int b_cs;
    float* out; int out_cs, C; long px;
};
VFI_HD void add_clamp_body(const AddClampArgs& a, long idx) {
    if (idx >= a.px * a.C) return;
    const long p = idx / a.C;
    const int c = (int)(idx - p * a.C);
    const float v = a.a[p * a.a_cs + c] + a.b[p * a.b_cs + c];
    a.out[p * a.out_cs + c] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
}

// ---- ResynNet's blend (:188-192), cropped: softmax over (clamp(m0), clamp(m1), 0) weights img0, img1, deg ----------------------
struct ResynBlendArgs {
    const float* img0; const float* img1; const float* deg; int img_cs;
    const float* m0; const float* m1; int m_cs;
    float* out; int Hp, Wp, H, W;
};
VFI_HD void resyn_blend_body(const ResynBlendArgs& a, long idx) {
    if (idx >= (long)a.H * a.W) return;
    const int x = (int)(idx % a.W), y = (int)(idx / a.W);
    const size_t p = (size_t)y * a.Wp + x;
    float l0 = a.m0[p * a.m_cs], l1 = a.m1[p * a.m_cs];
    l0 = l0 < -4.f ? -4.f : (l0 > 4.f ? 4.f : l0);
    l1 = l1 < -4.f ? -4.f : (l1 > 4.f ? 4.f : l1);
    float mx = l0 > l1 ? l0 : l1;
    mx = mx > 0.f ? mx : 0.f;
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx), e2 = expf(0.f - mx);
    const float sum = e0 + e1 + e2;
    for (int c = 0; c < 3; ++c) {
        float v = 0.f;
        v += a.img0[p * a.img_cs + c] * (e0 / sum);
        v += a.img1[p * a.img_cs + c] * (e1 / sum);
        v += a.deg[p * a.img_cs + c] * (e2 / sum);
        a.out[idx * 3 + c] = v;
    }
}

// ---- a constant over a channel window (timestep planes) ---------------------------------------------------------------------
struct FillArgs {
    float* out; int cs, C; long px; float v;
};
VFI_HD void fill_body(const FillArgs& a, long idx) {
    if (idx >= a.px * a.C) return;
    const long p = idx / a.C;
    a.out[p * a.cs + (idx - p * a.C)] = a.v;
}

}  // namespace vfi_ifunet
