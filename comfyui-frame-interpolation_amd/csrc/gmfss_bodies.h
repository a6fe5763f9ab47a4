// Per-element bodies of the GMFSS Fortuna kernels.  Every kernel of gmfss_ops.hip is
//     __global__ k(Args a) { body(a, blockIdx.x * blockDim.x + threadIdx.x); }
// with the body defined here as a __host__ __device__ function, so the same source also runs element by element on the
// host: tests/hostcheck builds these bodies into a host-side test library (hipcc, no GPU needed) and the CPU test suite
// checks each one against torch before anything reaches an MI355X.  First-correct versions: one thread per output
// element with plain loops (the attention / matching products are VALU loops, not MFMA tiles yet).
// Reference: vfi_models/gmfss_fortuna/GMFSS_Fortuna_union_arch.py (line numbers below refer to it).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>

#ifndef VFI_HD
#define VFI_HD __host__ __device__ static inline
#endif

namespace vfi_gmfss {

// ---- F.pad of an RGB frame into channels 0..2 of an NHWC tensor (gmfss_fortuna/__init__.py:43-48) --------------------
struct PadRgbArgs {
    const float* frame; int C, H, W;
    float* out; int out_cs, Hp, Wp;
};
VFI_HD void pad_rgb_body(const PadRgbArgs& a, long idx) {
    if (idx >= (long)a.Hp * a.Wp) return;
    const int x = (int)(idx % a.Wp), y = (int)(idx / a.Wp);
    float* o = a.out + (size_t)idx * a.out_cs;
    const bool in = y < a.H && x < a.W;
    const float* p = a.frame + ((size_t)(in ? y : 0) * a.W + (in ? x : 0)) * a.C;
    for (int c = 0; c < 3; ++c) o[c] = in ? p[c] : 0.f;
}

// ---- (x - mean_c) / std_c: normalize_img (:1123-1131) -----------------------------------------------------------------
struct NormChanArgs {
    const float* in; int in_cs;
    float* out; int out_cs, C; long px;
    float mean[8], std[8];
};
VFI_HD void norm_chan_body(const NormChanArgs& a, long idx) {
    if (idx >= a.px * a.C) return;
    const long p = idx / a.C;
    const int c = (int)(idx - p * a.C);
    a.out[p * a.out_cs + c] = (a.in[p * a.in_cs + c] - a.mean[c]) / a.std[c];
}

// ---- nn.PReLU() with one shared slope, over a channel window ------------------------------------------------------------
struct PreluArgs {
    const float* in; int in_cs;
    float* out; int out_cs, C; long px;
    float slope;
};
VFI_HD void prelu_body(const PreluArgs& a, long idx) {
    if (idx >= a.px * a.C) return;
    const long p = idx / a.C;
    const int c = (int)(idx - p * a.C);
    const float v = a.in[p * a.in_cs + c];
    a.out[p * a.out_cs + c] = v > 0.f ? v : v * a.slope;
}

// ---- InstanceNorm2d (no affine, eps 1e-5, biased variance), :165-215,232-234 -------------------------------------------
// pass 1: partial sums per (n, strip, c) in double; pass 2: mean and 1/sqrt(var+eps) per (n, c)
struct InStatsArgs {
    const float* x; int cs, C, N; long HW; int strips;
    double* part;    // [N][strips][C][2]
};
VFI_HD void instnorm_partial_body(const InStatsArgs& a, long idx) {
    if (idx >= (long)a.N * a.strips * a.C) return;
    const int c = (int)(idx % a.C);
    const int s = (int)((idx / a.C) % a.strips);
    const int n = (int)(idx / ((long)a.C * a.strips));
    const long per = (a.HW + a.strips - 1) / a.strips, lo = s * per, hi = lo + per < a.HW ? lo + per : a.HW;
    const float* b = a.x + (size_t)n * a.HW * a.cs + c;
    double s1 = 0.0, s2 = 0.0;
    for (long p = lo; p < hi; ++p) {
        const double v = b[p * a.cs];
        s1 += v;
        s2 += v * v;
    }
    a.part[idx * 2] = s1;
    a.part[idx * 2 + 1] = s2;
}
struct InFinalArgs {
    const double* part; int C, N; long HW; int strips;
    float* stats;    // [N][C][2] = mean, rstd
    float eps;
};
VFI_HD void instnorm_final_body(const InFinalArgs& a, long idx) {
    if (idx >= (long)a.N * a.C) return;
    const int c = (int)(idx % a.C), n = (int)(idx / a.C);
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < a.strips; ++s) {
        const double* q = a.part + (((size_t)n * a.strips + s) * a.C + c) * 2;
        s1 += q[0];
        s2 += q[1];
    }
    const double mean = s1 / (double)a.HW;
    double var = s2 / (double)a.HW - mean * mean;
    if (var < 0.0) var = 0.0;
    a.stats[idx * 2] = (float)mean;
    a.stats[idx * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
}
// out = act2( act1((x - mean) * rstd) + add )
struct InApplyArgs {
    const float* x; int cs; const float* stats; int C, N; long HW;
    int relu1; const float* add; int add_cs, relu2;
    float* out; int out_cs;
};
VFI_HD void instnorm_apply_body(const InApplyArgs& a, long idx) {
    if (idx >= (long)a.N * a.HW * a.C) return;
    const int c = (int)(idx % a.C);
    const long p = idx / a.C;            // n*HW + pixel
    const int n = (int)(p / a.HW);
    const float* st = a.stats + ((size_t)n * a.C + c) * 2;
    float v = (a.x[p * a.cs + c] - st[0]) * st[1];
    if (a.relu1) v = v > 0.f ? v : 0.f;
    if (a.add) v += a.add[p * a.add_cs + c];
    if (a.relu2) v = v > 0.f ? v : 0.f;
    a.out[p * a.out_cs + c] = v;
}

// ---- nn.LayerNorm(C) over the channel axis of [tokens, C] (eps 1e-5), :479-523 -----------------------------------------
struct LayerNormArgs {
    const float* x; int cs, C; long tokens;
    const float* gamma; const float* beta;
    float* out; int out_cs; float eps;
    const float* add; int add_cs;       // nullable: out = add + LN(x)   (source + message: TransformerLayer.forward's residual, :523)
    float* out2; int out2_cs;           // nullable: a second copy of the result (the FFN's concat slot)
};
VFI_HD void layernorm_body(const LayerNormArgs& a, long idx) {
    if (idx >= a.tokens) return;
    const float* b = a.x + idx * a.cs;
    float mean = 0.f;
    for (int c = 0; c < a.C; ++c) mean += b[c];
    mean /= (float)a.C;
    float var = 0.f;
    for (int c = 0; c < a.C; ++c) {
        const float d = b[c] - mean;
        var += d * d;
    }
    const float rstd = 1.0f / sqrtf(var / (float)a.C + a.eps);
    float* o = a.out + idx * a.out_cs;
    for (int c = 0; c < a.C; ++c) {
        float v = (b[c] - mean) * rstd * a.gamma[c] + a.beta[c];
        if (a.add) v += a.add[idx * a.add_cs + c];
        o[c] = v;
        if (a.out2) a.out2[idx * a.out2_cs + c] = v;
    }
}

// ---- nn.GELU() (erf form) in place over a channel window -----------------------------------------------------------------
struct GeluArgs {
    float* x; int cs, C; long px;
};
VFI_HD void gelu_body(const GeluArgs& a, long idx) {
    if (idx >= a.px * a.C) return;
    const long p = idx / a.C;
    float* q = a.x + p * a.cs + (idx - p * a.C);
    const float v = *q;
    *q = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

// ---- roll + split into attention windows, and back (:367-436, split_feature / merge_splits :1059-1120) ----------------
// forward: win[(b*K + wy)*K + wx][ly*ww + lx][c] = img[b][(wy*wh + ly + sh) % h][(wx*ww + lx + sw) % w][c]
// inverse: the same index map, written the other way round
struct WinPartArgs {
    const float* in; int in_cs;
    float* out; int out_cs;
    int B, h, w, C, K, sh, sw, inverse;
};
VFI_HD void window_partition_body(const WinPartArgs& a, long idx) {
    if (idx >= (long)a.B * a.h * a.w * a.C) return;
    const int c = (int)(idx % a.C);
    long t = idx / a.C;
    const int wh = a.h / a.K, ww = a.w / a.K;
    const int l = (int)(t % (wh * ww));
    t /= (wh * ww);
    const int wx = (int)(t % a.K), wy = (int)((t / a.K) % a.K), b = (int)(t / ((long)a.K * a.K));
    const int ly = l / ww, lx = l - ly * ww;
    const int Y = (wy * wh + ly + a.sh) % a.h, X = (wx * ww + lx + a.sw) % a.w;
    const size_t img = ((size_t)b * a.h + Y) * a.w + X;
    const size_t win = (((size_t)b * a.K + wy) * a.K + wx) * (wh * ww) + l;
    if (a.inverse) a.out[img * a.out_cs + c] = a.in[win * a.in_cs + c];
    else a.out[win * a.out_cs + c] = a.in[img * a.in_cs + c];
}

// ---- batched products of the attention / matching steps ---------------------------------------------------------------------
// out[b][m][n] = alpha * sum_k A[b][m][k] * Bm[b][n][k]        (q k^T / sqrt(c), :319,417-420,810-812)
struct BmmNtArgs {
    const float* A; int a_cs; const float* Bm; int b_cs;
    float* out; int nb, M, N, K; float alpha;
};
VFI_HD void bmm_nt_body(const BmmNtArgs& a, long idx) {
    if (idx >= (long)a.nb * a.M * a.N) return;
    const int n = (int)(idx % a.N);
    const int m = (int)((idx / a.N) % a.M);
    const int b = (int)(idx / ((long)a.N * a.M));
    const float* p = a.A + ((size_t)b * a.M + m) * a.a_cs;
    const float* q = a.Bm + ((size_t)b * a.N + n) * a.b_cs;
    float s = 0.f;
    for (int k = 0; k < a.K; ++k) s += p[k] * q[k];
    a.out[idx] = s * a.alpha;
}
// the same product with a 4x4 output tile per thread and float4 loads along K (16 FMA per pair of 4+4 vector loads instead
// of 1 per 2 scalar loads); needs K % 4 == 0 and 16-byte aligned rows.  Rows past M / N are clamped on load, skipped on store.
VFI_HD void bmm_nt4_body(const BmmNtArgs& a, long idx) {
    const int Mt = (a.M + 3) / 4, Nt = (a.N + 3) / 4;
    if (idx >= (long)a.nb * Mt * Nt) return;
    const int nt = (int)(idx % Nt), mt = (int)((idx / Nt) % Mt), b = (int)(idx / ((long)Nt * Mt));
    const float* A = a.A + (size_t)b * a.M * a.a_cs;
    const float* B = a.Bm + (size_t)b * a.N * a.b_cs;
    const float* ar[4];
    const float* br[4];
    for (int i = 0; i < 4; ++i) {
        const int m = 4 * mt + i < a.M ? 4 * mt + i : a.M - 1, n = 4 * nt + i < a.N ? 4 * nt + i : a.N - 1;
        ar[i] = A + (size_t)m * a.a_cs;
        br[i] = B + (size_t)n * a.b_cs;
    }
    float acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k = 0; k < a.K; k += 4) {
        float4 av[4], bv[4];
        for (int i = 0; i < 4; ++i) {
            av[i] = *(const float4*)(ar[i] + k);
            bv[i] = *(const float4*)(br[i] + k);
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                acc[i][j] += ((av[i].x * bv[j].x + av[i].y * bv[j].y) + av[i].z * bv[j].z) + av[i].w * bv[j].w;
    }
    for (int i = 0; i < 4; ++i) {
        const int m = 4 * mt + i;
        if (m >= a.M) break;
        float* o = a.out + ((size_t)b * a.M + m) * a.N;
        for (int j = 0; j < 4; ++j)
            if (4 * nt + j < a.N) o[4 * nt + j] = acc[i][j] * a.alpha;
    }
}
// out[b][m][c] = sum_n P[b][m][n] * V[b][n][c]                 (attn v, prob grid, prob flow)
struct BmmNnArgs {
    const float* P; const float* V; int v_cs;
    float* out; int out_cs, nb, M, N, C;
};
VFI_HD void bmm_nn_body(const BmmNnArgs& a, long idx) {
    if (idx >= (long)a.nb * a.M * a.C) return;
    const int c = (int)(idx % a.C);
    const int m = (int)((idx / a.C) % a.M);
    const int b = (int)(idx / ((long)a.C * a.M));
    const float* p = a.P + ((size_t)b * a.M + m) * a.N;
    const float* v = a.V + (size_t)b * a.N * a.v_cs + c;
    float s = 0.f;
    for (int n = 0; n < a.N; ++n) s += p[n] * v[(size_t)n * a.v_cs];
    a.out[((size_t)b * a.M + m) * a.out_cs + c] = s;
}
// softmax over the rows of x [nb][rows][cols] in place; mask (nullable) [period][rows][cols] is added first, batch b
// taking mask[b % period] (scores += attn_mask.repeat(b, 1, 1), :422-423)
struct SoftmaxArgs {
    float* x; int nb, rows, cols; const float* mask; int period;
};
VFI_HD void softmax_rows_body(const SoftmaxArgs& a, long idx) {
    if (idx >= (long)a.nb * a.rows) return;
    float* r = a.x + idx * a.cols;
    const int b = (int)(idx / a.rows), row = (int)(idx % a.rows);
    const float* mk = a.mask ? a.mask + ((size_t)(b % a.period) * a.rows + row) * a.cols : nullptr;
    float mx = -INFINITY;
    for (int j = 0; j < a.cols; ++j) {
        const float v = mk ? r[j] + mk[j] : r[j];
        r[j] = v;
        mx = v > mx ? v : mx;
    }
    float sum = 0.f;
    for (int j = 0; j < a.cols; ++j) {
        const float e = expf(r[j] - mx);
        r[j] = e;
        sum += e;
    }
    const float inv = 1.0f / sum;
    for (int j = 0; j < a.cols; ++j) r[j] *= inv;
}

// ---- bilinear_sample / flow_warp (:955-991): zeros padding, align_corners=True, at pixel + flow ---------------------------
struct ZTap {
    int x0, y0; float wx, wy;     // taps (x0,y0),(x0+1,y0),(x0,y0+1),(x0+1,y0+1); out-of-image taps contribute 0
};
VFI_HD ZTap ztap_from_norm(float gx, float gy, int W, int H) {
    // grid_sample un-normalisation, align_corners=True: ((g + 1) / 2) * (size - 1)
    const float px = ((gx + 1.0f) / 2.0f) * (float)(W - 1), py = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float fx = floorf(px), fy = floorf(py);
    ZTap t;
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    t.wx = px - fx;
    t.wy = py - fy;
    return t;
}
VFI_HD float ztap_read(const float* img, int cs, int W, int H, const ZTap& t, int c) {
    // torch: nw = (x1 - x) * (y1 - y) ... with x1 = x0 + 1; out-of-bounds taps are skipped
    const float e = 1.0f - t.wx, s = 1.0f - t.wy;
    float v = 0.f;
    const bool x0 = t.x0 >= 0 && t.x0 < W, x1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    const bool y0 = t.y0 >= 0 && t.y0 < H, y1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
    if (x0 && y0) v += img[((size_t)t.y0 * W + t.x0) * cs + c] * (e * s);
    if (x1 && y0) v += img[((size_t)t.y0 * W + t.x0 + 1) * cs + c] * (t.wx * s);
    if (x0 && y1) v += img[((size_t)(t.y0 + 1) * W + t.x0) * cs + c] * (e * t.wy);
    if (x1 && y1) v += img[((size_t)(t.y0 + 1) * W + t.x0 + 1) * cs + c] * (t.wx * t.wy);
    return v;
}
struct FlowSampleArgs {
    const float* in; int in_cs; const float* flow; int flow_cs;
    float* out; int out_cs, N, H, W, C;
};
VFI_HD void flow_sample_body(const FlowSampleArgs& a, long idx) {
    if (idx >= (long)a.N * a.H * a.W) return;
    const int X = (int)(idx % a.W), Y = (int)((idx / a.W) % a.H);
    const int n = (int)(idx / ((long)a.W * a.H));
    const float gx = 2.0f * ((float)X + a.flow[idx * a.flow_cs]) / (float)(a.W - 1) - 1.0f;
    const float gy = 2.0f * ((float)Y + a.flow[idx * a.flow_cs + 1]) / (float)(a.H - 1) - 1.0f;
    const ZTap t = ztap_from_norm(gx, gy, a.W, a.H);
    const float* img = a.in + (size_t)n * a.H * a.W * a.in_cs;
    float* o = a.out + (size_t)idx * a.out_cs;
    for (int c = 0; c < a.C; ++c) o[c] = ztap_read(img, a.in_cs, a.W, a.H, t, c);
}

// ---- F.interpolate(x, size, mode="bilinear", align_corners=True) * post_mul (:1302-1307) ------------------------------------
struct ResizeAcArgs {
    const float* in; int in_cs;
    float* out; int out_cs, N, Hi, Wi, Ho, Wo, C; float post_mul;
};
VFI_HD void resize_ac_body(const ResizeAcArgs& a, long idx) {
    if (idx >= (long)a.N * a.Ho * a.Wo) return;
    const int x = (int)(idx % a.Wo), y = (int)((idx / a.Wo) % a.Ho);
    const int n = (int)(idx / ((long)a.Wo * a.Ho));
    // area_pixel_compute_scale(align_corners=True) = (in - 1) / (out - 1); source index = scale * dst
    const float ry = a.Ho > 1 ? (float)(a.Hi - 1) / (float)(a.Ho - 1) : 0.f, rx = a.Wo > 1 ? (float)(a.Wi - 1) / (float)(a.Wo - 1) : 0.f;
    const float sy = ry * (float)y, sx = rx * (float)x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < a.Hi - 1 ? 1 : 0), x1 = x0 + (x0 < a.Wi - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const float* b = a.in + (size_t)n * a.Hi * a.Wi * a.in_cs;
    float* o = a.out + (size_t)idx * a.out_cs;
    for (int c = 0; c < a.C; ++c) {
        const float v00 = b[((size_t)y0 * a.Wi + x0) * a.in_cs + c], v01 = b[((size_t)y0 * a.Wi + x1) * a.in_cs + c];
        const float v10 = b[((size_t)y1 * a.Wi + x0) * a.in_cs + c], v11 = b[((size_t)y1 * a.Wi + x1) * a.in_cs + c];
        o[c] = (hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11)) * a.post_mul;
    }
}

// ---- local_correlation_softmax (:846-913): flow delta = E[window coordinate] - pixel, softmax over (2r+1)^2 --------------
struct LocalMatchArgs {
    const float* f0; int f0_cs; const float* f1; int f1_cs;
    float* flow; int flow_cs;    // flow[p, 0:2] += predicted residual (flow = flow + flow_pred, :1335)
    int N, H, W, C, R;
};
VFI_HD void local_match_body(const LocalMatchArgs& a, long idx) {
    if (idx >= (long)a.N * a.H * a.W) return;
    const int X = (int)(idx % a.W), Y = (int)((idx / a.W) % a.H);
    const int n = (int)(idx / ((long)a.W * a.H));
    const int D = 2 * a.R + 1;
    const float* q = a.f0 + (size_t)idx * a.f0_cs;
    const float* img = a.f1 + (size_t)n * a.H * a.W * a.f1_cs;
    const float cx = (float)(a.W - 1) / 2.0f, cy = (float)(a.H - 1) / 2.0f;
    const float scale = sqrtf((float)a.C);
    float corr[121];
    float mx = -INFINITY;
    for (int j = 0; j < D * D; ++j) {
        // window_grid: offsets enumerated row-major, (dx, dy) = (j % D - R, j / D - R) (:935-946,866-876)
        const float sx = (float)X + (float)(j % D - a.R), sy = (float)Y + (float)(j / D - a.R);
        const bool valid = sx >= 0.f && sx < (float)a.W && sy >= 0.f && sy < (float)a.H;
        const ZTap t = ztap_from_norm((sx - cx) / cx, (sy - cy) / cy, a.W, a.H);
        float s = 0.f;
        for (int c = 0; c < a.C; ++c) s += q[c] * ztap_read(img, a.f1_cs, a.W, a.H, t, c);
        s = valid ? s / scale : -1e9f;
        corr[j] = s;
        mx = s > mx ? s : mx;
    }
    float sum = 0.f;
    for (int j = 0; j < D * D; ++j) {
        corr[j] = expf(corr[j] - mx);
        sum += corr[j];
    }
    float ex = 0.f, ey = 0.f;
    for (int j = 0; j < D * D; ++j) {
        const float p = corr[j] / sum;
        ex += p * ((float)X + (float)(j % D - a.R));
        ey += p * ((float)Y + (float)(j / D - a.R));
    }
    a.flow[idx * a.flow_cs] += ex - (float)X;
    a.flow[idx * a.flow_cs + 1] += ey - (float)Y;
}

// ---- FeatureFlowAttention.forward_local_window_attn (:745-803): 3x3 (radius R) window, zero padding as unfold's -------------
struct LocalPropArgs {
    const float* q; int q_cs; const float* k; int k_cs; const float* flow; int flow_cs;
    float* out; int out_cs, N, H, W, C, R;
};
VFI_HD void local_prop_body(const LocalPropArgs& a, long idx) {
    if (idx >= (long)a.N * a.H * a.W) return;
    const int X = (int)(idx % a.W), Y = (int)((idx / a.W) % a.H);
    const int n = (int)(idx / ((long)a.W * a.H));
    const int D = 2 * a.R + 1;
    const float* q = a.q + (size_t)idx * a.q_cs;
    const float scale = sqrtf((float)a.C);
    float sc[49], fx[49], fy[49];
    float mx = -INFINITY;
    for (int j = 0; j < D * D; ++j) {
        const int x = X + j % D - a.R, y = Y + j / D - a.R;
        const bool in = x >= 0 && x < a.W && y >= 0 && y < a.H;
        float s = 0.f;
        fx[j] = fy[j] = 0.f;
        if (in) {
            const size_t p = ((size_t)n * a.H + y) * a.W + x;
            const float* kk = a.k + p * a.k_cs;
            for (int c = 0; c < a.C; ++c) s += q[c] * kk[c];
            fx[j] = a.flow[p * a.flow_cs];
            fy[j] = a.flow[p * a.flow_cs + 1];
        }
        s /= scale;          // padded taps: key = 0 -> score 0, value 0; they DO take part in the softmax
        sc[j] = s;
        mx = s > mx ? s : mx;
    }
    float sum = 0.f;
    for (int j = 0; j < D * D; ++j) {
        sc[j] = expf(sc[j] - mx);
        sum += sc[j];
    }
    float ox = 0.f, oy = 0.f;
    for (int j = 0; j < D * D; ++j) {
        const float p = sc[j] / sum;
        ox += p * fx[j];
        oy += p * fy[j];
    }
    a.out[idx * a.out_cs] = ox;
    a.out[idx * a.out_cs + 1] = oy;
}

// ---- GMFlow.upsample_flow, convex branch (:1237-1258): K x K fine pixels per coarse pixel ---------------------------------
struct ConvexUpArgs {
    const float* mask; int mask_cs;     // [N,H,W, 9*K*K], channel = (j*K + ky)*K + kx
    const float* flow; int flow_cs;     // [N,H,W,2]
    float* out; int out_cs;             // [N,K*H,K*W,2]
    int N, H, W, K;
};
VFI_HD void convex_up_body(const ConvexUpArgs& a, long idx) {
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
    float ox = 0.f, oy = 0.f;
    for (int j = 0; j < 9; ++j) {
        const int x = X + j % 3 - 1, y = Y + j / 3 - 1;       // F.unfold(K * flow, [3,3], padding=1): zero outside
        if (x < 0 || x >= a.W || y < 0 || y >= a.H) continue;
        const float* f = a.flow + (((size_t)n * a.H + y) * a.W + x) * a.flow_cs;
        const float pw = w[j] / sum;
        ox += pw * ((float)a.K * f[0]);
        oy += pw * ((float)a.K * f[1]);
    }
    float* o = a.out + (((size_t)n * a.H * a.K + (size_t)Y * a.K + ky) * ((size_t)a.W * a.K) + (size_t)X * a.K + kx) * a.out_cs;
    o[0] = ox;
    o[1] = oy;
}

// ---- MetricNet input assembly (:1375-1454): 14 channels = img0 3, img1 3, -l1(img0, backwarp(img1, f01)), ---------------
// -l1(img1, backwarp(img0, f10)), f01 / ((size-1)/2), f10 / ((size-1)/2), fwd occlusion, bwd occlusion
struct MetricInArgs {
    const float* img0; const float* img1; int img_cs;      // [H,W,>=3]
    const float* f01; const float* f10; int f_cs;          // [H,W,2]
    float* out; int out_cs, H, W;
};
VFI_HD float lin11h(int i, int n) {
    // torch.linspace(-1, 1, n)[i] in fp32: start + step*i below the midpoint, end - step*(n-1-i) above it
    const float step = 2.0f / (float)(n - 1);
    return i < n / 2 ? -1.0f + step * (float)i : 1.0f - step * (float)(n - 1 - i);
}
VFI_HD void metric_inputs_body(const MetricInArgs& a, long idx) {
    if (idx >= (long)a.H * a.W) return;
    const int X = (int)(idx % a.W), Y = (int)(idx / a.W);
    const float hw = ((float)a.W - 1.0f) / 2.0f, hh = ((float)a.H - 1.0f) / 2.0f;
    const float* fa = a.f01 + idx * a.f_cs;
    const float* fb = a.f10 + idx * a.f_cs;
    float* o = a.out + idx * a.out_cs;
    // backwarp(): grid = linspace + flow / ((size-1)/2), zeros padding
    const ZTap ta = ztap_from_norm(lin11h(X, a.W) + fa[0] / hw, lin11h(Y, a.H) + fa[1] / hh, a.W, a.H);
    const ZTap tb = ztap_from_norm(lin11h(X, a.W) + fb[0] / hw, lin11h(Y, a.H) + fb[1] / hh, a.W, a.H);
    float m0 = 0.f, m1 = 0.f;
    for (int c = 0; c < 3; ++c) {
        const float i0 = a.img0[idx * a.img_cs + c], i1 = a.img1[idx * a.img_cs + c];
        o[c] = i0;
        o[3 + c] = i1;
        m0 += fabsf(i0 - ztap_read(a.img1, a.img_cs, a.W, a.H, ta, c));
        m1 += fabsf(i1 - ztap_read(a.img0, a.img_cs, a.W, a.H, tb, c));
    }
    o[6] = -(m0 / 3.0f);
    o[7] = -(m1 / 3.0f);
    o[8] = fa[0] / hw;
    o[9] = fa[1] / hh;
    o[10] = fb[0] / hw;
    o[11] = fb[1] / hh;
    // forward_backward_consistency_check (:994-1012) with flow_warp's pixel-grid sampling
    const float mag = sqrtf(fa[0] * fa[0] + fa[1] * fa[1]) + sqrtf(fb[0] * fb[0] + fb[1] * fb[1]);
    const ZTap sa = ztap_from_norm(2.0f * ((float)X + fa[0]) / (float)(a.W - 1) - 1.0f, 2.0f * ((float)Y + fa[1]) / (float)(a.H - 1) - 1.0f, a.W, a.H);
    const ZTap sb = ztap_from_norm(2.0f * ((float)X + fb[0]) / (float)(a.W - 1) - 1.0f, 2.0f * ((float)Y + fb[1]) / (float)(a.H - 1) - 1.0f, a.W, a.H);
    const float wbx = ztap_read(a.f10, a.f_cs, a.W, a.H, sa, 0), wby = ztap_read(a.f10, a.f_cs, a.W, a.H, sa, 1);   // bwd flow at x + fwd
    const float wfx = ztap_read(a.f01, a.f_cs, a.W, a.H, sb, 0), wfy = ztap_read(a.f01, a.f_cs, a.W, a.H, sb, 1);   // fwd flow at x + bwd
    const float df = sqrtf((fa[0] + wbx) * (fa[0] + wbx) + (fa[1] + wby) * (fa[1] + wby));
    const float db = sqrtf((fb[0] + wfx) * (fb[0] + wfx) + (fb[1] + wfy) * (fb[1] + wfy));
    const float thr = 0.01f * mag + 0.5f;
    o[12] = df > thr ? 1.0f : 0.0f;
    o[13] = db > thr ? 1.0f : 0.0f;
}

// ---- tanh(x) * s over a channel window (:1465) -----------------------------------------------------------------------------
struct TanhArgs {
    float* x; int cs, C; long px; float s;
};
VFI_HD void tanh_scale_body(const TanhArgs& a, long idx) {
    if (idx >= a.px * a.C) return;
    const long p = idx / a.C;
    float* q = a.x + p * a.cs + (idx - p * a.C);
    *q = tanhf(*q) * a.s;
}

// ---- softsplat(x, flow, metric, "soft") around the summation splat (cupy_ops/softsplat.py:408-432) ---------------------------
// prep: out[p] = (x[p, 0:C] * exp(zs * z[p]), exp(zs * z[p])), flow_out[p] = fs * flow[p]   (Z_t = t * metric, F_t = t * flow)
struct SplatPrepArgs {
    const float* x; int x_cs; const float* z; int z_cs; const float* flow; int flow_cs;
    float* out; float* flow_out;      // [px, C+1] and [px, 2], dense
    int C; long px; float zs, fs;
};
VFI_HD void splat_prep_body(const SplatPrepArgs& a, long idx) {
    if (idx >= a.px * (a.C + 1)) return;
    const long p = idx / (a.C + 1);
    const int c = (int)(idx - p * (a.C + 1));
    const float e = expf(a.zs * a.z[p * a.z_cs]);
    a.out[idx] = c < a.C ? a.x[p * a.x_cs + c] * e : e;
    if (c == 0) {
        a.flow_out[p * 2] = a.fs * a.flow[p * a.flow_cs];
        a.flow_out[p * 2 + 1] = a.fs * a.flow[p * a.flow_cs + 1];
    }
}
// norm: out[p, 0:C] = s[p, 0:C] / (s[p, C] + 1e-7)
struct SplatNormArgs {
    const float* s; float* out; int out_cs, C; long px;
};
VFI_HD void splat_norm_body(const SplatNormArgs& a, long idx) {
    if (idx >= a.px * a.C) return;
    const long p = idx / a.C;
    const int c = (int)(idx - p * a.C);
    a.out[p * a.out_cs + c] = a.s[p * (a.C + 1) + c] / (a.s[p * (a.C + 1) + a.C] + 0.0000001f);
}

// ---- nn.PixelShuffle(2): out[2y+dy, 2x+dx, c] = in[y, x, 4c + 2dy + dx] -------------------------------------------------
struct PixShufArgs {
    const float* in; int in_cs;
    float* out; int out_cs, N, H, W, C;       // C = output channels; in has 4C
};
VFI_HD void pixel_shuffle2_body(const PixShufArgs& a, long idx) {
    if (idx >= (long)a.N * 4 * a.H * a.W * a.C) return;
    const int c = (int)(idx % a.C);
    const long p = idx / a.C;
    const int X = (int)(p % (2 * a.W)), Y = (int)((p / (2 * a.W)) % (2 * a.H));
    const int n = (int)(p / ((long)4 * a.W * a.H));
    const size_t src = ((size_t)n * a.H + Y / 2) * a.W + X / 2;
    a.out[p * a.out_cs + c] = a.in[src * a.in_cs + 4 * c + 2 * (Y & 1) + (X & 1)];
}

// ---- torch.clamp(out, 0, 1)[:, :, :H, :W] -----------------------------------------------------------------------------------
struct ClampCropArgs {
    const float* in; int in_cs, Hp, Wp;
    float* out; int H, W, C;
};
VFI_HD void clamp_crop_body(const ClampCropArgs& a, long idx) {
    if (idx >= (long)a.H * a.W * a.C) return;
    const int c = (int)(idx % a.C);
    const long p = idx / a.C;
    const int x = (int)(p % a.W), y = (int)(p / a.W);
    const float v = a.in[((size_t)y * a.Wp + x) * a.in_cs + c];
    a.out[idx] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
}

}  // namespace vfi_gmfss
