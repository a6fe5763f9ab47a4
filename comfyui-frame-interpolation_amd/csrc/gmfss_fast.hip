// Device forms of three steps of GMFSS's `prepare` that the per-element bodies of gmfss_bodies.h leave slow on an MI355X
// (round-2 profile at 1080p: layernorm 8.7 ms, local_match 15.0 ms, instnorm_partial 4.4 ms of a 70 ms pair):
//
//   layernorm_wave_kernel   nn.LayerNorm(128) of FeatureTransformer (GMFSS_Fortuna_union_arch.py:479-523).  The body walks a
//                           512-byte token row per THREAD (64 rows per load instruction); here one wave owns a token, the row
//                           is one coalesced read, mean / variance are shuffle reductions.  HBM-bound: 2 x 4 B per element.
//
//   instnorm_partial_wg_kernel  the first pass of nn.InstanceNorm2d (:165-215), a workgroup per strip (see below).
//
//   local_match_mfma_kernel local_correlation_softmax (:846-913) with C = 128, radius 4: for every pixel, softmax over the
//                           9 x 9 window of  q . bilinear_sample(f1, window position) / sqrt(C), expected window coordinate.
//                           The sampling positions go through grid_sample's normalise / un-normalise round trip, so they are
//                           integers only up to float rounding: a tap is a blend of up to four neighbours with weights
//                           (1-e)(1-e') ...; the blend is linear in f1, so  q . sample = sum_taps w_tap * (q . f1[tap])  and
//                           the integer-position dot products D are a banded GEMM.  One wave = a 4 x 8 tile of queries;
//                           its keys are the 14 x 18 patch of f1 around the tile (window +-4, +1 for the blend partner),
//                           8 blocks of 32 keys; D^T = K Q^T on v_mfma_f32_32x32x2_f32 (exact fp32 products) with the query
//                           as the lane's column, K rows read straight from L2 into the A operand (double-buffered in
//                           registers; the 8.4 MB feature map is L2 / MALL resident), D parked in LDS [query][key]; then each
//                           query's two lanes evaluate the 81 taps exactly as the body does (same ZTap arithmetic, same
//                           validity rule, -1e9 for windows positions outside the image) with an online softmax.
//                           81 of 256 products per query are used (0.32): 8.6 GFLOP per 1080p call instead of the body's 81 x
//                           128 x 4 uncoalesced reads per pixel.
#include "gmfss_fast.h"

#include "vfi_common.h"

namespace vfi {

using vfi_gmfss::LayerNormArgs;
using vfi_gmfss::LocalMatchArgs;
using vfi_gmfss::ZTap;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------------- layer norm
namespace {

constexpr int LN_MAXC = 256;

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void layernorm_wave_kernel(const LayerNormArgs a) {
    const long tok = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tok >= a.tokens) return;
    const float* b = a.x + tok * a.cs;
    float v[LN_MAXC / 64];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC / 64; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < a.C ? b[c] : 0.f;
        s += v[i];
    }
    const float mean = wave_sum(s) / (float)a.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXC / 64; ++i) {
        const float d = lane + 64 * i < a.C ? v[i] - mean : 0.f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)a.C + a.eps);
    float* o = a.out + tok * a.out_cs;
#pragma unroll
    for (int i = 0; i < LN_MAXC / 64; ++i) {
        const int c = lane + 64 * i;
        if (c < a.C) {
            float r = (v[i] - mean) * rstd * a.gamma[c] + a.beta[c];
            if (a.add) r += a.add[tok * a.add_cs + c];
            o[c] = r;
            if (a.out2) a.out2[tok * a.out2_cs + c] = r;
        }
    }
}

}  // namespace

bool layernorm_wave_fits(const LayerNormArgs& a) { return a.C <= LN_MAXC; }

int layernorm_wave_launch(const LayerNormArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("layernorm", s);
    hipLaunchKernelGGL(layernorm_wave_kernel, dim3((unsigned)((a.tokens + 3) / 4)), dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ instance-norm sums
// instnorm_partial_body gives each (image, strip, channel) to ONE thread: N x strips x C = 8192 threads for the backbone's
// [2, 544, 960, 64] maps, 128 waves on 1024 SIMDs, each walking 8160 pixels serially (0.29 ms per call, 15 calls per pair).
// Here a workgroup owns the strip: 256 / C pixel lanes x C channels, so a wave reads whole 256-byte pixel rows; the same
// double-precision partial sums land in the same [N][strips][C][2] workspace for instnorm_final_body.
namespace {

__global__ __launch_bounds__(256) void instnorm_partial_wg_kernel(const vfi_gmfss::InStatsArgs a) {
    __shared__ double sm[2][256];
    const int s = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int P = 256 / a.C;
    const int p = tid / a.C, c = tid - p * a.C;
    const long per = (a.HW + a.strips - 1) / a.strips, lo = s * per, hi = lo + per < a.HW ? lo + per : a.HW;
    double s1 = 0.0, s2 = 0.0;
    if (p < P) {
        const float* b = a.x + (size_t)n * a.HW * a.cs + c;
#pragma unroll 8
        for (long i = lo + p; i < hi; i += P) {
            const double v = b[i * a.cs];
            s1 += v;
            s2 += v * v;
        }
    }
    sm[0][tid] = s1, sm[1][tid] = s2;
    __syncthreads();
    if (tid < a.C) {
        for (int k = 1; k < P; ++k) s1 += sm[0][tid + k * a.C], s2 += sm[1][tid + k * a.C];
        double* q = a.part + (((size_t)n * a.strips + s) * a.C + tid) * 2;
        q[0] = s1, q[1] = s2;
    }
}

// second pass: one wave per (image, channel) sums the strips' partials (the body: one thread walks all of them)
__global__ __launch_bounds__(256) void instnorm_final_wave_kernel(const vfi_gmfss::InFinalArgs a) {
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (idx >= (long)a.N * a.C) return;
    const int c = (int)(idx % a.C), n = (int)(idx / a.C);
    double s1 = 0.0, s2 = 0.0;
    for (int s = lane; s < a.strips; s += 64) {
        const double* q = a.part + (((size_t)n * a.strips + s) * a.C + c) * 2;
        s1 += q[0];
        s2 += q[1];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o), s2 += __shfl_xor(s2, o);
    if (lane == 0) {
        const double mean = s1 / (double)a.HW;
        double var = s2 / (double)a.HW - mean * mean;
        if (var < 0.0) var = 0.0;
        a.stats[idx * 2] = (float)mean;
        a.stats[idx * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
}

}  // namespace

bool instnorm_partial_wg_fits(const vfi_gmfss::InStatsArgs& a) { return a.C <= 256; }

int instnorm_final_wave_launch(const vfi_gmfss::InFinalArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("instnorm_final", s);
    hipLaunchKernelGGL(instnorm_final_wave_kernel, dim3((unsigned)(((long)a.N * a.C + 3) / 4)), dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

int instnorm_partial_wg_launch(const vfi_gmfss::InStatsArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("instnorm_partial", s);
    hipLaunchKernelGGL(instnorm_partial_wg_kernel, dim3(a.strips, a.N), dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

// --------------------------------------------------------------------------------------------------------- local match
namespace {

constexpr int LM_C = 128, LM_R = 4, LM_D = 2 * LM_R + 1;
constexpr int LM_TH = 4, LM_TW = 8;                       // query tile (32 queries = one MFMA column block)
constexpr int LM_B = LM_R + 1;                            // patch border: window radius + the blend partner
constexpr int LM_PH = LM_TH + 2 * LM_B, LM_PW = LM_TW + 2 * LM_B;   // 14 x 18 keys
constexpr int LM_KEYS = 256;                              // 8 blocks of 32 (252 used)
constexpr int LM_SS = LM_KEYS + 1;                        // LDS row stride of D[query][key]
static_assert(LM_PH * LM_PW <= LM_KEYS, "patch does not fit the key blocks");

__global__ __launch_bounds__(64) void local_match_mfma_kernel(const LocalMatchArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float D[32 * LM_SS];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int n = blockIdx.z;
    const int Y0 = blockIdx.y * LM_TH, X0 = blockIdx.x * LM_TW;
    const int qy = l31 >> 3, qx = l31 & 7;
    const int Y = Y0 + qy, X = X0 + qx;
    const bool qok = Y < a.H && X < a.W;
    const float* img = a.f1 + (size_t)n * a.H * a.W * a.f1_cs;

    // B operand: this lane's query, channels [64 half, 64 half + 64)   (k is only a summation index: any channel order that
    // A and B share is valid, and this one makes every lane's operand stream one contiguous 256-byte run)
    float qreg[64];
    {
        const float* qp = a.f0 + ((size_t)(n * a.H + (qok ? Y : 0)) * a.W + (qok ? X : 0)) * a.f0_cs + 64 * half;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            f32x4 t = *(const f32x4*)(qp + 4 * g);
            if (!qok) t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) qreg[4 * g + j] = t[j];
        }
    }
    // A operand of key block mb: key 32 mb + l31 of the patch, same channel half
    f32x4 kr[2][16];
    auto fetch = [&](int mb, f32x4* dst) {
        const int kidx = 32 * mb + l31;
        const int py = kidx / LM_PW, px = kidx - py * LM_PW;
        const int ky = Y0 - LM_B + py, kx = X0 - LM_B + px;
        const bool in = py < LM_PH && ky >= 0 && ky < a.H && kx >= 0 && kx < a.W;
        const float* kp = img + ((size_t)(in ? ky : 0) * a.W + (in ? kx : 0)) * a.f1_cs + 64 * half;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            f32x4 t = *(const f32x4*)(kp + 4 * g);
            dst[g] = in ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(0, kr[0]);
#pragma unroll
    for (int mb = 0; mb < LM_KEYS / 32; ++mb) {
        if (mb + 1 < LM_KEYS / 32) fetch(mb + 1, kr[(mb + 1) & 1]);
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[mb & 1][g][j], qreg[4 * g + j], s, 0, 0, 0);
        // accumulator register r of half h = key row 8 (r / 4) + 4 h + r % 4, column = this lane's query
#pragma unroll
        for (int r = 0; r < 16; ++r) D[l31 * LM_SS + 32 * mb + 8 * (r >> 2) + 4 * half + (r & 3)] = s[r];
    }
    __syncthreads();

    // ---- the 81 window positions, split over the query's two lanes (41 + 40); online softmax
    const float cx = (float)(a.W - 1) / 2.0f, cy = (float)(a.H - 1) / 2.0f;
    const float scale = sqrtf((float)a.C);
    const float* Dq = D + l31 * LM_SS;
    float m_run = -INFINITY, l_run = 0.f, ex = 0.f, ey = 0.f;
    const int j0 = half ? 41 : 0, j1 = half ? LM_D * LM_D : 41;
    for (int j = j0; j < j1; ++j) {
        const int jy = j / LM_D, jx = j - jy * LM_D;
        const float sx = (float)X + (float)(jx - LM_R), sy = (float)Y + (float)(jy - LM_R);
        const bool valid = sx >= 0.f && sx < (float)a.W && sy >= 0.f && sy < (float)a.H;
        const ZTap t = vfi_gmfss::ztap_from_norm((sx - cx) / cx, (sy - cy) / cy, a.W, a.H);
        // ztap_read with D in place of the image: out-of-image taps are skipped
        const float e = 1.0f - t.wx, sw = 1.0f - t.wy;
        const bool bx0 = t.x0 >= 0 && t.x0 < a.W, bx1 = t.x0 + 1 >= 0 && t.x0 + 1 < a.W;
        const bool by0 = t.y0 >= 0 && t.y0 < a.H, by1 = t.y0 + 1 >= 0 && t.y0 + 1 < a.H;
        int px = t.x0 - (X0 - LM_B), py = t.y0 - (Y0 - LM_B);
        px = px < 0 ? 0 : (px > LM_PW - 2 ? LM_PW - 2 : px);     // (cannot leave the patch: |rounding| << 1 px)
        py = py < 0 ? 0 : (py > LM_PH - 2 ? LM_PH - 2 : py);
        const float* d = Dq + py * LM_PW + px;
        float s = 0.f;
        if (bx0 && by0) s += d[0] * (e * sw);
        if (bx1 && by0) s += d[1] * (t.wx * sw);
        if (bx0 && by1) s += d[LM_PW] * (e * t.wy);
        if (bx1 && by1) s += d[LM_PW + 1] * (t.wx * t.wy);
        s = valid ? s / scale : -1e9f;
        const float m_new = fmaxf(m_run, s);
        const float c = expf(m_run - m_new), p = expf(s - m_new);
        l_run = l_run * c + p;
        ex = ex * c + p * (float)(jx - LM_R);     // E[offset] = E[window coordinate] - pixel since sum p = 1: the body's
        ey = ey * c + p * (float)(jy - LM_R);     // sum p (X + dx) - X loses X * 2^-23 per term to cancellation, this does not
        m_run = m_new;
    }
    const float m_o = __shfl_xor(m_run, 32), l_o = __shfl_xor(l_run, 32), ex_o = __shfl_xor(ex, 32), ey_o = __shfl_xor(ey, 32);
    const float m_t = fmaxf(m_run, m_o);
    const float c0 = expf(m_run - m_t), c1 = expf(m_o - m_t);
    const float l_t = l_run * c0 + l_o * c1;
    if (qok && half == 0) {
        const size_t idx = (size_t)(n * a.H + Y) * a.W + X;
        a.flow[idx * a.flow_cs] += (ex * c0 + ex_o * c1) / l_t;
        a.flow[idx * a.flow_cs + 1] += (ey * c0 + ey_o * c1) / l_t;
    }
#endif
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- local propagation
// FeatureFlowAttention.forward_local_window_attn (:745-803), C = 128, radius 1: softmax over the 3 x 3 window of q . k / sqrt(C)
// (padded taps score 0 and carry a zero flow, as unfold's zero padding), weighted sum of the window's flows.  The body walks
// 9 x 128 strided floats per THREAD; here 16 lanes share a pixel (8 channels each, two float4 loads per tap) and the nine dot
// products are reduced across them with xor-shuffles.
namespace {

__global__ __launch_bounds__(256) void local_prop_coop_kernel(const vfi_gmfss::LocalPropArgs a) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long idx = gid >> 4;                 // pixel
    const int sub = (int)(gid & 15);           // channels [8 sub, 8 sub + 8)
    const bool ok = idx < (long)a.N * a.H * a.W;
    const long pid = ok ? idx : 0;
    const int X = (int)(pid % a.W), Y = (int)((pid / a.W) % a.H);
    const int n = (int)(pid / ((long)a.W * a.H));
    const float* q = a.q + (size_t)pid * a.q_cs + 8 * sub;
    const f32x4 q0 = *(const f32x4*)q, q1 = *(const f32x4*)(q + 4);
    const float scale = sqrtf((float)a.C);
    float sc[9], fx[9], fy[9];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int x = X + j % 3 - 1, y = Y + j / 3 - 1;
        const bool in = x >= 0 && x < a.W && y >= 0 && y < a.H;
        const size_t p = ((size_t)n * a.H + (in ? y : Y)) * a.W + (in ? x : X);
        const float* kk = a.k + p * a.k_cs + 8 * sub;
        const f32x4 k0 = *(const f32x4*)kk, k1 = *(const f32x4*)(kk + 4);
        float s = q0[0] * k0[0] + q0[1] * k0[1] + q0[2] * k0[2] + q0[3] * k0[3] + q1[0] * k1[0] + q1[1] * k1[1] + q1[2] * k1[2] + q1[3] * k1[3];
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        s = in ? s / scale : 0.f;          // padded taps: key = 0 -> score 0, value 0; they DO take part in the softmax
        fx[j] = in ? a.flow[p * a.flow_cs] : 0.f;
        fy[j] = in ? a.flow[p * a.flow_cs + 1] : 0.f;
        sc[j] = s;
        mx = s > mx ? s : mx;
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        sc[j] = expf(sc[j] - mx);
        sum += sc[j];
    }
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const float p = sc[j] / sum;
        ox += p * fx[j];
        oy += p * fy[j];
    }
    if (ok && sub == 0) {
        a.out[idx * a.out_cs] = ox;
        a.out[idx * a.out_cs + 1] = oy;
    }
}

}  // namespace

bool local_prop_coop_fits(const vfi_gmfss::LocalPropArgs& a) {
    return a.C == 128 && a.R == 1 && a.q_cs % 4 == 0 && a.k_cs % 4 == 0 && (((uintptr_t)a.q | (uintptr_t)a.k) & 15) == 0;
}

int local_prop_coop_launch(const vfi_gmfss::LocalPropArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("local_propagate", s);
    const long threads = (long)a.N * a.H * a.W * 16;
    hipLaunchKernelGGL(local_prop_coop_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

bool local_match_mfma_fits(const LocalMatchArgs& a) {
    return a.C == LM_C && a.R == LM_R && a.f0_cs % 4 == 0 && a.f1_cs % 4 == 0 && (((uintptr_t)a.f0 | (uintptr_t)a.f1) & 15) == 0;
}

int local_match_mfma_launch(const LocalMatchArgs& a, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    TraceScope ts("local_match", s);
    hipLaunchKernelGGL(local_match_mfma_kernel, dim3(cdiv(a.W, LM_TW), cdiv(a.H, LM_TH), a.N), dim3(64), 0, s, a);
    VFI_CHECK_HIP(hipGetLastError());
    return 0;
}

}  // namespace vfi
