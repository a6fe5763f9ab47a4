// Flash-style single-head attention on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 products) for GMFSS's
// GMFlow: the 12 + 12 window attentions of FeatureTransformer, the global matching softmax (8640 x 8640 at 1080p) and the
// global flow propagation.  Replaces the three-kernel form  scores = q k^T (bmm_nt)  ->  row softmax  ->  P v (bmm_nn)  of
// round 1, whose score matrices round-tripped HBM (11 GiB of workspace at 1080p, 215 of the 275 ms per pair):
//   GMFSS_Fortuna_union_arch.py:367-436 (single_head_split_window_attention), :806-843 (global_correlation_softmax),
//   :708-745 (FeatureFlowAttention.forward, global branch).
//
// out[b][m][0:DV] = sum_n softmax_n(alpha * q[b][m] . k[b][n] + mask(b, m, n)) * v[b][n][0:DV]
//   mask(b, m, n) = -100 where label[b % period][m] != label[b % period][n]  (the shifted-window mask, :326-364), else 0.
//
// One wave = 32 queries, one workgroup = 4 waves = 128 queries of one batch entry; keys / values stream through LDS in blocks
// of 32 (double-buffered).  The transposed products are computed, S^T = K Q^T and O^T = V^T P^T, so that a lane's MFMA
// column is ONE query: the online-softmax row statistics are per-lane scalars (plus one exchange with lane ^ 32), the
// rescaling of the output accumulator is a per-lane multiply, and the probabilities never leave registers — the S^T
// accumulator layout (register r of half h = key 8(r/4) + 4h + r%4) is exactly the B-operand order of the second product when
// V^T is read from LDS four keys at a time.
#include <map>
#include <mutex>
#include <utility>

#include "vfi_common.h"

#include "../../include/vfi_hip.h"

namespace vfi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ATT_C = 128;            // head dimension of GMFlow's transformer
constexpr int ATT_KB = 32;            // keys per block
constexpr int ATT_KS = ATT_C + 4;     // LDS row stride of the K tile (floats)
constexpr int ATT_VS = ATT_KB + 4;    // LDS row stride of the V^T tile

struct AttArgs {
    const float* q;
    const float* k;
    const float* v;
    float* out;
    const int* labels;   // [period][L] or nullptr
    int q_cs, k_cs, v_cs, out_cs;
    int nb, Lq, Lk, DV, period;
    float alpha;
    // window mode (win_K > 0): q / k / v / out are [B, h, w, .] token MAPS and batch entry b = (image, wy, wx) is the K x K
    // window of the map rolled by (-sh, -sw): token l of the window = pixel ((wy wh + l / ww + sh) % h, (wx ww + l % ww + sw) % w)
    // — torch.roll + split_feature + merge_splits + roll back (:367-436, :1059-1120) folded into the addressing.
    int win_K, win_h, win_w, win_sh, win_sw, win_wh, win_ww;   // wh x ww = the window
    // key split (set by the launcher, see attention_launch_t): blockIdx.z owns a range of key blocks and leaves its un-normalised
    // output rows and softmax statistics (running maximum, sum) in a workspace; attention_merge_kernel combines the splits
    int ksplit;
    float* part_o;    // [ksplit][nb][Lq][DV]
    float* part_ml;   // [ksplit][nb][Lq][2]
    float win_inv_ww;
};

// row (token index into the [.., cs] arrays) of token l of batch entry b
__device__ inline size_t att_row(const AttArgs& a, int b, int l, int L, int wy0, int wx0, int img0) {
    if (!a.win_K) return (size_t)b * L + l;
    const int ww = a.win_ww;
    const int ly = (int)(((float)l + 0.5f) * a.win_inv_ww);     // exact: l < 2^20, the quotient is >= 0.5 / ww from an integer
    const int lx = l - ly * ww;
    int y = wy0 + ly, x = wx0 + lx;                              // wy0 = wy wh + sh < 2h
    y = y >= a.win_h ? y - a.win_h : y;
    x = x >= a.win_w ? x - a.win_w : x;
    return (size_t)(img0 + y) * a.win_w + x;
}

template <int DVT>   // output tiles of 32 value channels: 4 (DV = 128) or 1 (DV <= 32, e.g. the 2-channel grid / flow)
__global__ __launch_bounds__(256) void attention_kernel(const AttArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int DVP = DVT * 32;
    constexpr int KT = ATT_KB * ATT_KS, VT = DVP * ATT_VS;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x (K tile | V^T tile | labels)
    constexpr int BUF = KT + VT + ATT_KB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int query = q0 + l31;
    const bool qok = query < a.Lq;
    int wy0 = 0, wx0 = 0, img0 = 0;
    if (a.win_K) {
        const int K = a.win_K, wimg = b / (K * K), wy = (b / K) % K, wx = b % K;
        wy0 = wy * a.win_wh + a.win_sh, wx0 = wx * a.win_ww + a.win_sw, img0 = wimg * a.win_h;
    }
    const int* lab = a.labels ? a.labels + (size_t)(b % a.period) * a.Lk : nullptr;     // (labels index tokens: Lq == Lk when used)

    // this lane's query row, the channels of its half: qreg[4g + j] = Q[query][8g + 4 half + j]
    float qreg[64];
    {
        const float* qp = a.q + att_row(a, b, qok ? query : 0, a.Lq, wy0, wx0, img0) * a.q_cs + 4 * half;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            f32x4 t = *(const f32x4*)(qp + 8 * g);
            if (!qok) t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) qreg[4 * g + j] = t[j];
        }
    }
    const int qlab = lab && qok ? lab[query] : 0;

    f32x16 o[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int nblk = (a.Lk + ATT_KB - 1) / ATT_KB;
    // loader, split in two so that the global latency hides under the MFMAs: fetch() block k+1 into registers before block k is
    // multiplied, commit() them to the other LDS buffer afterwards.  K tile 32 x 128 floats = 1024 float4 (4 per thread),
    // V tile 32 keys x DV (DV = 128: 4 float4 per thread, transposed on the way into LDS).
    f32x4 kreg[4], vreg[4];
    int lreg = 0;
    auto fetch_k = [&](int kblk) {
        const int key0 = kblk * ATT_KB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int key = idx >> 5, d4 = idx & 31;
            kreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (key0 + key < a.Lk) kreg[i] = *(const f32x4*)(a.k + att_row(a, b, key0 + key, a.Lk, wy0, wx0, img0) * a.k_cs + 4 * d4);
        }
    };
    auto fetch_v = [&](int kblk) {
        const int key0 = kblk * ATT_KB;
        if (DVT == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                const int key = idx & 31, d4 = idx >> 5;     // consecutive lanes = consecutive keys: conflict-free transposed store
                vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (key0 + key < a.Lk) vreg[i] = *(const f32x4*)(a.v + att_row(a, b, key0 + key, a.Lk, wy0, wx0, img0) * a.v_cs + 4 * d4);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                const int key = idx & 31, d = idx >> 5;
                vreg[0][i] = (d < a.DV && key0 + key < a.Lk) ? a.v[att_row(a, b, key0 + key, a.Lk, wy0, wx0, img0) * a.v_cs + d] : 0.f;
            }
        }
        if (tid < ATT_KB) lreg = (lab && key0 + tid < a.Lk) ? lab[key0 + tid] : 0;
    };
    auto commit_k = [&](int buf) {
        float* kt = smem + buf * BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int key = idx >> 5, d4 = idx & 31;
            *(f32x4*)(kt + key * ATT_KS + 4 * d4) = kreg[i];
        }
    };
    auto commit_v = [&](int buf) {
        float* vt = smem + buf * BUF + KT;
        int* lt = (int*)(vt + VT);
        if (DVT == 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                const int key = idx & 31, d4 = idx >> 5;
#pragma unroll
                for (int j = 0; j < 4; ++j) vt[(4 * d4 + j) * ATT_VS + key] = vreg[i][j];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                vt[(idx >> 5) * ATT_VS + (idx & 31)] = vreg[0][i];
            }
        }
        if (tid < ATT_KB) lt[tid] = lreg;
    };

    // S^T = K Q^T : rows = keys, this lane's column = its query.  Two accumulator chains (even / odd channel groups), summed at the
    // end: 64 matrix instructions into ONE accumulator are 64 dependent issues, each waiting for the previous result to clear
    // the pipe (the convolution kernels rotate their accumulators for the same reason, conv_mfma2.hip)
    auto qk = [&](const float* kt) {
        f32x16 s0, s1;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f, s1[r] = 0.f;
        const float* krow = kt + l31 * ATT_KS + 4 * half;
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
            const f32x4 ka = *(const f32x4*)(krow + 8 * g), kb = *(const f32x4*)(krow + 8 * g + 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[j], qreg[4 * g + j], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[j], qreg[4 * g + 4 + j], s1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] += s1[r];
        return s0;
    };
    // scale, mask, online softmax of block kblk (per-lane: every register of `s` belongs to this lane's query), then
    // O^T += V^T P^T : A = V^T[channel l31 of tile t][4 keys of this half], B = the S^T registers as they are
    auto softmax_pv = [&](f32x16 s, int kblk, const float* vt, const int* lt) {
        const int key0 = kblk * ATT_KB;
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = 8 * (r >> 2) + 4 * half + (r & 3);
            float x = s[r] * a.alpha;
            x += lt[kk] != qlab ? -100.0f : 0.0f;        // branch-free: without labels the tile and qlab are all 0
            x = key0 + kk >= a.Lk ? -INFINITY : x;
            s[r] = x;
            mloc = fmaxf(mloc, x);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);          // finite: every block holds at least one real key
        const float corr = __expf(m_run - m_new);        // 0 on the first block (m_run = -inf)
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __expf(s[r] - m_new);
            s[r] = p;
            psum += p;
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= corr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {              // consecutive matrix instructions rotate through the DVT output tiles
            f32x4 vf[DVT];
#pragma unroll
            for (int t = 0; t < DVT; ++t) vf[t] = *(const f32x4*)(vt + (32 * t + l31) * ATT_VS + 4 * half + 8 * j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < DVT; ++t) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[t][i], s[4 * j + i], o[t], 0, 0, 0);
        }
    };

    int kb0 = 0, kb1 = nblk;
    if (a.ksplit > 1) {
        const int per = (nblk + a.ksplit - 1) / a.ksplit;
        kb0 = blockIdx.z * per;
        kb1 = kb0 + per < nblk ? kb0 + per : nblk;      // (the launcher leaves no split empty)
    }
    fetch_k(kb0);
    fetch_v(kb0);
    commit_k(0);
    commit_v(0);
    __syncthreads();
    for (int kblk = kb0; kblk < kb1; ++kblk) {
        const int buf = (kblk - kb0) & 1;
        if (kblk + 1 < kb1) fetch_k(kblk + 1), fetch_v(kblk + 1);
        const float* kt = smem + buf * BUF;
        const float* vt = kt + KT;
        softmax_pv(qk(kt), kblk, vt, (const int*)(vt + VT));
        if (kblk + 1 < kb1) commit_k(buf ^ 1), commit_v(buf ^ 1);      // that buffer was released by the barrier that ended block kblk-1
        __syncthreads();                                               // block consumed; the next one has been written
    }
    // ---- normalise and store: register r of tile t = channel 32t + 8(r/4) + 4 half + r%4 of this lane's query
    if (qok && a.ksplit > 1) {      // partial result of this key range: raw accumulator + (m, l)
        const size_t row = ((size_t)blockIdx.z * a.nb + b) * a.Lq + query;
        float* op = a.part_o + row * a.DV;
        if (half == 0) a.part_ml[2 * row] = m_run, a.part_ml[2 * row + 1] = l_run;
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int ch = 32 * t + 8 * r4 + 4 * half;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ch + i < a.DV) op[ch + i] = o[t][4 * r4 + i];
            }
    } else if (qok) {
        const float inv = 1.0f / l_run;
        float* op = a.out + att_row(a, b, query, a.Lq, wy0, wx0, img0) * a.out_cs;
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int ch = 32 * t + 8 * r4 + 4 * half;
                if (DVT == 4) {
                    *(f32x4*)(op + ch) = f32x4{o[t][4 * r4] * inv, o[t][4 * r4 + 1] * inv, o[t][4 * r4 + 2] * inv, o[t][4 * r4 + 3] * inv};
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ch + i < a.DV) op[ch + i] = o[t][4 * r4 + i] * inv;
                }
            }
    }
#endif
}

// out[b][query][c] = sum_z o_z e^(m_z - M) / sum_z l_z e^(m_z - M),  M = max_z m_z: the online-softmax merge of the key splits
__global__ __launch_bounds__(256) void attention_merge_kernel(const AttArgs a) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)a.nb * a.Lq * a.DV;
    if (idx >= total) return;
    const long row = idx / a.DV;
    const int c = (int)(idx - row * a.DV);
    const int b = (int)(row / a.Lq), query = (int)(row - (long)b * a.Lq);
    const long rows = (long)a.nb * a.Lq;
    float M = -INFINITY;
    for (int z = 0; z < a.ksplit; ++z) M = fmaxf(M, a.part_ml[2 * (z * rows + row)]);
    float num = 0.f, den = 0.f;
    for (int z = 0; z < a.ksplit; ++z) {
        const float w = __expf(a.part_ml[2 * (z * rows + row)] - M);
        num += a.part_o[(z * rows + row) * a.DV + c] * w;
        den += a.part_ml[2 * (z * rows + row) + 1] * w;
    }
    int wy0 = 0, wx0 = 0, img0 = 0;
    if (a.win_K) {
        const int K = a.win_K, wimg = b / (K * K), wy = (b / K) % K, wx = b % K;
        wy0 = wy * a.win_wh + a.win_sh, wx0 = wx * a.win_ww + a.win_sw, img0 = wimg * a.win_h;
    }
    a.out[att_row(a, b, query, a.Lq, wy0, wx0, img0) * a.out_cs + c] = num / den;
}

// split workspace per (device, stream), grow-only
static int attention_workspace(int dev, hipStream_t s, size_t floats, float** out) {
    struct Ws {
        float* p = nullptr;
        size_t n = 0;
    };
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, Ws> table;
    std::lock_guard<std::mutex> lock(mu);
    Ws& w = table[{dev, s}];
    if (w.n < floats) {
        // the outgrown block is RETIRED, not freed: a captured HIP graph of the stream's owner may have its address baked in (r6), and
        // hipFree would drain the device under the other pair lanes
        w.p = nullptr, w.n = 0;
        VFI_CHECK_HIP(hipMalloc((void**)&w.p, floats * sizeof(float)));
        w.n = floats;
    }
    *out = w.p;
    return 0;
}

template <int DVT>
static int attention_launch_t(const AttArgs& a_in, hipStream_t s) {
    AttArgs a = a_in;
    constexpr int LDS = 2 * (ATT_KB * ATT_KS + DVT * 32 * ATT_VS + ATT_KB) * 4;
    static bool attr_set[kMaxDevices] = {};
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "attention: device index %d out of range", dev);
    if (!attr_set[dev]) {
        VFI_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attention_kernel<DVT>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set[dev] = true;
    }
    // Key split: GMFlow's coarse scale has 8 windows x 16 query blocks = 128 workgroups for 256 CUs (global matching: 2 x 64), each
    // walking 64 ... 255 key blocks.  Then the keys are cut into `ks` ranges (gridDim.z) and a merge kernel combines the partial
    // softmaxes; ks fills the CUs once (this kernel holds one workgroup per CU) and leaves >= 8 key blocks per range.
    static int cus_of[kMaxDevices] = {};
    if (!cus_of[dev]) {
        hipDeviceProp_t p;
        VFI_CHECK_HIP(hipGetDeviceProperties(&p, dev));
        cus_of[dev] = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
    }
    const int wgs = cdiv(a.Lq, 128) * a.nb, nblk = cdiv(a.Lk, ATT_KB);
    int ks = 1;
    if (wgs < cus_of[dev] && nblk >= 16) {
        ks = (DVT == 4 ? 1 : 2) * cus_of[dev] / wgs;      // (the 2-channel form fits two workgroups per CU)
        ks = ks > nblk / 8 ? nblk / 8 : ks;
        ks = ks > 8 ? 8 : ks;
        if (ks > 1) {
            const int per = cdiv(nblk, ks);
            ks = cdiv(nblk, per);          // no empty range
        }
    }
    a.ksplit = ks > 1 ? ks : 0, a.part_o = nullptr, a.part_ml = nullptr;
    if (ks > 1) {
        const size_t rows = (size_t)ks * a.nb * a.Lq;
        float* ws = nullptr;
        if (int rc = attention_workspace(dev, s, rows * (a.DV + 2), &ws)) return rc;
        a.part_o = ws, a.part_ml = ws + rows * a.DV;
    }
    {
        TraceScope ts(DVT == 4 ? "attention_c128" : "attention_c2", s);
        hipLaunchKernelGGL(attention_kernel<DVT>, dim3(cdiv(a.Lq, 128), a.nb, ks > 1 ? ks : 1), dim3(256), LDS, s, a);
        VFI_CHECK_HIP(hipGetLastError());
    }
    if (ks > 1) {
        TraceScope ts("attention_merge", s);
        const long total = (long)a.nb * a.Lq * a.DV;
        hipLaunchKernelGGL(attention_merge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
        VFI_CHECK_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_attention(const float* q_dev, int q_cs, const float* k_dev, int k_cs, const float* v_dev, int v_cs, float* out_dev,
                             int out_cs, int nb, int Lq, int Lk, int C, int DV, float alpha, const int* labels_dev, int label_period,
                             void* stream) {
    VFI_REQUIRE(q_dev && k_dev && v_dev && out_dev && nb > 0 && Lq > 0 && Lk > 0, "vfi_attention: bad arguments");
    VFI_REQUIRE(C == ATT_C, "vfi_attention: head dimension %d (only %d, GMFlow's)", C, ATT_C);
    VFI_REQUIRE(DV == 128 || (DV >= 1 && DV <= 32), "vfi_attention: %d value channels (128, or 1..32)", DV);
    VFI_REQUIRE(q_cs >= C && k_cs >= C && q_cs % 4 == 0 && k_cs % 4 == 0 && v_cs >= DV && out_cs >= DV &&
                    (DV != 128 || (v_cs % 4 == 0 && out_cs % 4 == 0)),
                "vfi_attention: bad strides (q %d, k %d, v %d, out %d)", q_cs, k_cs, v_cs, out_cs);
    VFI_REQUIRE((((uintptr_t)q_dev | (uintptr_t)k_dev) & 15) == 0 && (DV != 128 || (((uintptr_t)v_dev | (uintptr_t)out_dev) & 15) == 0),
                "vfi_attention: unaligned pointers");
    VFI_REQUIRE(!labels_dev || (label_period > 0 && Lq == Lk), "vfi_attention: labels need Lq == Lk and a period");
    AttArgs a;
    a.q = q_dev, a.k = k_dev, a.v = v_dev, a.out = out_dev, a.labels = labels_dev;
    a.q_cs = q_cs, a.k_cs = k_cs, a.v_cs = v_cs, a.out_cs = out_cs;
    a.nb = nb, a.Lq = Lq, a.Lk = Lk, a.DV = DV, a.period = labels_dev ? label_period : 1;
    a.alpha = alpha;
    a.win_K = 0, a.win_h = a.win_w = a.win_sh = a.win_sw = a.win_wh = a.win_ww = 0, a.win_inv_ww = 0.f;
    return DV == 128 ? attention_launch_t<4>(a, (hipStream_t)stream) : attention_launch_t<1>(a, (hipStream_t)stream);
}

extern "C" int vfi_window_attention(const float* q_dev, int q_cs, const float* k_dev, int k_cs, const float* v_dev, int v_cs, float* out_dev,
                                    int out_cs, int B, int h, int w, int splits, int shift_h, int shift_w, int C, float alpha,
                                    const int* labels_dev, void* stream) {
    VFI_REQUIRE(q_dev && k_dev && v_dev && out_dev && B > 0 && h > 0 && w > 0 && splits > 0 && h % splits == 0 && w % splits == 0,
                "vfi_window_attention: bad arguments (%dx%d into %d splits)", h, w, splits);
    VFI_REQUIRE(C == ATT_C, "vfi_window_attention: head dimension %d (only %d, GMFlow's)", C, ATT_C);
    VFI_REQUIRE(shift_h >= 0 && shift_h < h / splits && shift_w >= 0 && shift_w < w / splits, "vfi_window_attention: shift (%d,%d) outside a window", shift_h, shift_w);
    VFI_REQUIRE(q_cs >= C && k_cs >= C && v_cs >= C && out_cs >= C && (q_cs | k_cs | v_cs | out_cs) % 4 == 0, "vfi_window_attention: bad strides");
    VFI_REQUIRE((((uintptr_t)q_dev | (uintptr_t)k_dev | (uintptr_t)v_dev | (uintptr_t)out_dev) & 15) == 0, "vfi_window_attention: unaligned pointers");
    VFI_REQUIRE(out_dev != q_dev && out_dev != k_dev && out_dev != v_dev, "vfi_window_attention: out aliases an input");
    VFI_REQUIRE((long)(h / splits) * (w / splits) < (1 << 20), "vfi_window_attention: window too large");
    AttArgs a;
    a.q = q_dev, a.k = k_dev, a.v = v_dev, a.out = out_dev, a.labels = labels_dev;
    a.q_cs = q_cs, a.k_cs = k_cs, a.v_cs = v_cs, a.out_cs = out_cs;
    a.nb = B * splits * splits, a.Lq = a.Lk = (h / splits) * (w / splits), a.DV = C, a.period = splits * splits;
    a.alpha = alpha;
    a.win_K = splits, a.win_h = h, a.win_w = w, a.win_sh = shift_h, a.win_sw = shift_w, a.win_wh = h / splits, a.win_ww = w / splits, a.win_inv_ww = 1.0f / (float)(w / splits);
    return attention_launch_t<4>(a, (hipStream_t)stream);
}
