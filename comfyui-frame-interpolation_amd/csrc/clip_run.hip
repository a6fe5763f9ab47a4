// vfi_film_run / vfi_m2m_run — the whole FILM / M2M node call for a HOST clip behind one C entry point each ("same
// triplet" as RIFE's vfi_rife_run), for host applications that do not want to re-implement the node loops on top of
// vfi_film_forward / vfi_m2m_prepare + _render.  Built only on the public C ABI of include/vfi_hip.h plus the HIP runtime.
//
//   FILM  (vfi_models/film/__init__.py:63-113): per kept pair the greedy bisection of :12-42 (every call asks the model for the
//         midpoint of two already known frames; later calls consume earlier OUTPUTS, clamped to [0,1]); a skipped pair is DROPPED
//         from the output, frame included (:89-90); the clip's last frame is appended (:106).  An int multiplier applies to every
//         pair, a list is padded with 2 (:84-87).
//   M2M   (vfi_models/m2m/__init__.py:33-60 -> vfi_utils.generic_frame_loop, vfi_utils.py:149-389, timestep mode): int multiplier:
//         frame_i, its m-1 new frames (none when the pair is skipped), ..., last frame.  List multiplier (padded with 2): every pair
//         runs as its own 2-frame loop — m == 0 drops the pair INCLUDING its first frame (and the clip's last frame when it is the
//         last pair), m == 1 keeps the frame, and the skip list is consulted with the pair's LOCAL index 0 (:364-386).
// Synchronous, one stream, pageable host buffers: simple and correct; both models are device-bound (60 ms per FILM frame, 10 ms
// per M2M pair at 1080p), so the host side is not where their time goes (the Python nodes overlap the copies anyway).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/vfi_hip.h"
#include "../../include/vfi_hip_test.h"
#include "vfi_common.h"

using namespace vfi;

namespace {

struct DevFrame {
    float* p = nullptr;
    ~DevFrame() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t floats) {
        if (p) (void)hipFree(p);
        p = nullptr;
        VFI_CHECK_HIP(hipMalloc((void**)&p, floats * sizeof(float)));
        return 0;
    }
};

void copy_rgb(float* dst, const float* src, size_t px, int C) {
    if (C == 3) {
        memcpy(dst, src, px * 3 * sizeof(float));
        return;
    }
    for (size_t i = 0; i < px; ++i) {
        dst[3 * i] = src[(size_t)C * i];
        dst[3 * i + 1] = src[(size_t)C * i + 1];
        dst[3 * i + 2] = src[(size_t)C * i + 2];
    }
}

// torch.linspace(0, 1, n) in float32 as torch's CPU kernel computes it (ATen RangeFactoriesKernel: first half start + step * i,
// second half end - step * (n - 1 - i), step = 1 / (n - 1) in float) — in a translation unit built with FMA, where the compiler
// contracts the second half into ONE rounding, fma(-step, k, end).  Checked value for value against torch.linspace for n = 2..199
// (tests/test_film_schedule.py); the bisection's exact ties (first at 9 frames per pair) are decided by these last bits.
std::vector<float> linspace01(int n) {
    std::vector<float> v(n);
    if (n == 1) {
        v[0] = 0.f;
        return v;
    }
    const float step = 1.0f / (float)(n - 1);
    const int halfway = n / 2;
    for (int i = 0; i < n; ++i) v[i] = i < halfway ? step * (float)i : fmaf(-step, (float)(n - 1 - i), 1.0f);
    return v;
}

}  // namespace

// film/__init__.py:17-40 — the order of (left, right, new) grid positions of one pair with `inter` new frames; fp32 arithmetic and
// first-minimum tie breaking as torch's (argmin over the row-major [interval][remaining] matrix)
static void film_schedule(int inter, std::vector<int>& triples) {
    triples.clear();
    std::vector<int> idxes = {0, inter + 1}, remains;
    for (int i = 1; i <= inter; ++i) remains.push_back(i);
    const std::vector<float> splits = linspace01(inter + 2);
    while (!remains.empty()) {
        float best = INFINITY;
        int bi = 0, bj = 0;
        bool first = true;
        for (size_t i = 0; i + 1 < idxes.size(); ++i) {
            const float start = splits[idxes[i]], end = splits[idxes[i + 1]];
            for (size_t j = 0; j < remains.size(); ++j) {
                const float d = fabsf((splits[remains[j]] - start) / (end - start) - 0.5f);
                if (first || d < best || (std::isnan(d) && !std::isnan(best))) {   // torch.argmin: first minimum; NaN counts as the minimum
                    best = d, bi = (int)i, bj = (int)j, first = false;
                }
            }
        }
        const int nw = remains[bj];
        triples.push_back(idxes[bi]);
        triples.push_back(idxes[bi + 1]);
        triples.push_back(nw);
        size_t pos = 0;
        while (pos < idxes.size() && idxes[pos] < nw) ++pos;     // bisect_left
        idxes.insert(idxes.begin() + pos, nw);
        remains.erase(remains.begin() + bj);
    }
}

extern "C" {

#ifdef VFI_TEST_TAPS      // include/vfi_hip_test.h: only in libvfi_hip_test.so
int vfi_test_linspace01(int n, float* out) {
    if (n < 1 || !out) {
        set_error("vfi_test_linspace01: bad arguments");
        return -1;
    }
    const std::vector<float> v = linspace01(n);
    memcpy(out, v.data(), v.size() * sizeof(float));
    return n;
}

int vfi_test_film_schedule(int inter_frames, int* triples, int cap) {
    if (inter_frames < 0 || (!triples && cap > 0)) {
        set_error("vfi_test_film_schedule: bad arguments");
        return -1;
    }
    std::vector<int> t;
    film_schedule(inter_frames, t);
    if ((int)t.size() > cap) {
        set_error("vfi_test_film_schedule: buffer too small (need %d ints)", (int)t.size());
        return -1;
    }
    if (!t.empty()) memcpy(triples, t.data(), t.size() * sizeof(int));
    return (int)t.size() / 3;
}
#endif  // VFI_TEST_TAPS

int vfi_film_run(vfi_film_t* net, const float* frames_host, int N, int H, int W, int C, int multiplier, const int* multipliers,
                 int n_multipliers, const uint8_t* skip, float* out_host, int64_t* n_out) {
    VFI_REQUIRE(net && n_out && N >= 1 && H > 0 && W > 0 && C >= 3, "vfi_film_run: bad arguments (N=%d H=%d W=%d C=%d)", N, H, W, C);
    // inference(..., inter_frames = m - 1) (film/__init__.py:12-41): m in {-1, 0, 1} runs no iteration and the node emits frame_i
    // alone (relust[:-1]); m <= -2 fails in torch.linspace(0, 1, m + 1) — the same here.
    std::vector<int> ms(N > 1 ? N - 1 : 0, 2);
    for (int p = 0; p + 1 < N; ++p) ms[p] = multipliers ? (p < n_multipliers ? multipliers[p] : 2) : multiplier;
    int64_t rows = 1;
    for (int p = 0; p + 1 < N; ++p)
        if (!(skip && skip[p])) {
            VFI_REQUIRE(ms[p] >= -1, "vfi_film_run: multiplier %d of pair %d (the reference fails in torch.linspace for multipliers <= -2)", ms[p], p);
            if (ms[p] < 1) ms[p] = 1;
            rows += ms[p];
        }
    *n_out = rows;
    if (!out_host) return 0;
    VFI_REQUIRE(frames_host, "vfi_film_run: null frames");
    const size_t px = (size_t)H * W, fin = px * C, fout = px * 3;
    hipStream_t st = nullptr;
    VFI_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamDestroy(s); }
    } guard{st};
    DevFrame d0, d1;
    if (d0.alloc(fin) || d1.alloc(fin)) return -1;
    int64_t row = 0;
    std::vector<int> sched;
    for (int p = 0; p + 1 < N; ++p) {
        if (skip && skip[p]) continue;
        const int m = ms[p];
        VFI_CHECK_HIP(hipMemcpyAsync(d0.p, frames_host + (size_t)p * fin, fin * sizeof(float), hipMemcpyHostToDevice, st));
        VFI_CHECK_HIP(hipMemcpyAsync(d1.p, frames_host + (size_t)(p + 1) * fin, fin * sizeof(float), hipMemcpyHostToDevice, st));
        // grid position -> device frame and its channel count (originals C, results 3)
        std::vector<DevFrame> mids(m > 1 ? m - 1 : 0);
        std::vector<const float*> at(m + 1, nullptr);
        std::vector<int> ch(m + 1, 3);
        at[0] = d0.p, ch[0] = C, at[m] = d1.p, ch[m] = C;
        film_schedule(m - 1, sched);
        for (size_t k = 0; k + 2 < sched.size(); k += 3) {
            const int l = sched[k], r = sched[k + 1], nw = sched[k + 2];
            VFI_REQUIRE(at[l] && at[r] && nw >= 1 && nw < m && !at[nw], "vfi_film_run: internal schedule error");
            if (mids[nw - 1].alloc(fout)) return -1;
            // both inputs must have the same channel count: an original (C channels) next to a result (3) goes through a 3-channel view
            const float *a = at[l], *b = at[r];
            int cc = ch[l];
            DevFrame ta, tb;
            if (ch[l] != ch[r]) {
                const bool left_orig = ch[l] != 3;
                DevFrame& tmp = left_orig ? ta : tb;
                if (tmp.alloc(fout)) return -1;
                VFI_CHECK_HIP(hipMemcpy2DAsync(tmp.p, 3 * sizeof(float), left_orig ? a : b, (size_t)C * sizeof(float), 3 * sizeof(float), px,
                                               hipMemcpyDeviceToDevice, st));
                (left_orig ? a : b) = tmp.p;
                cc = 3;
            }
            if (int rc = vfi_film_forward(net, a, b, cc, H, W, mids[nw - 1].p, 1, st)) return rc;
            VFI_CHECK_HIP(hipStreamSynchronize(st));      // (the temporaries above die here)
            at[nw] = mids[nw - 1].p;
        }
        copy_rgb(out_host + (size_t)row++ * fout, frames_host + (size_t)p * fin, px, C);
        for (int k = 1; k < m; ++k)
            VFI_CHECK_HIP(hipMemcpyAsync(out_host + (size_t)row++ * fout, at[k], fout * sizeof(float), hipMemcpyDeviceToHost, st));
        VFI_CHECK_HIP(hipStreamSynchronize(st));
    }
    copy_rgb(out_host + (size_t)row++ * fout, frames_host + (size_t)(N - 1) * fin, px, C);
    VFI_REQUIRE(row == rows, "vfi_film_run: internal row count error");
    return 0;
}

int vfi_m2m_run(vfi_m2m_t* net, const float* frames_host, int N, int H, int W, int C, int multiplier, const int* multipliers,
                int n_multipliers, const uint8_t* skip, float* out_host, int64_t* n_out) {
    VFI_REQUIRE(net && n_out && N >= 2 && H > 0 && W > 0 && C >= 3, "vfi_m2m_run: bad arguments (N=%d H=%d W=%d C=%d; the model needs >= 2 frames)", N,
                H, W, C);
    // ---- output plan (schedule.generic_output_plan): entries (frame index | -1, pair, timestep)
    struct Row {
        int src;      // >= 0: pass-through frame; -1: new frame
        int pair;
        float t;
    };
    std::vector<Row> plan;
    auto one_pair = [&](int pair, int m, bool skipped) {
        plan.push_back({pair, pair, 0.f});
        if (skipped || m <= 1) return;
        for (int k = 1; k < m; ++k) plan.push_back({-1, pair, (float)((double)k / (double)m)});     // python: k / m in double, then float32
    };
    if (!multipliers) {
        VFI_REQUIRE(multiplier >= 1, "vfi_m2m_run: multiplier %d", multiplier);
        for (int p = 0; p + 1 < N; ++p) one_pair(p, multiplier, skip && skip[p]);
        plan.push_back({N - 1, N - 1, 0.f});
    } else {
        for (int p = 0; p + 1 < N; ++p) {
            const int m = p < n_multipliers ? multipliers[p] : 2;
            if (m == 0) continue;
            // (the reference allocates torch.zeros(m * 2, ...) for the pair, vfi_utils.py:178: a negative m raises there)
            VFI_REQUIRE(m > 0, "vfi_m2m_run: multiplier %d of pair %d (the reference's frame loop fails on negative multipliers)", m, p);
            one_pair(p, m, skip && skip[0]);
            if (p == N - 2) plan.push_back({N - 1, N - 1, 0.f});
        }
    }
    *n_out = (int64_t)plan.size();
    if (!out_host) return 0;
    VFI_REQUIRE(frames_host, "vfi_m2m_run: null frames");
    const size_t px = (size_t)H * W, fin = px * C, fout = px * 3;
    hipStream_t st = nullptr;
    VFI_CHECK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    struct StreamGuard {
        hipStream_t s;
        ~StreamGuard() { (void)hipStreamDestroy(s); }
    } guard{st};
    DevFrame f[2], outd;
    if (f[0].alloc(fin) || f[1].alloc(fin) || outd.alloc(fout)) return -1;
    int have[2] = {-1, -1};      // frame index resident in f[0] / f[1]
    int prepared = -1;
    for (size_t i = 0; i < plan.size(); ++i) {
        const Row& r = plan[i];
        float* dst = out_host + i * fout;
        if (r.src >= 0) {
            copy_rgb(dst, frames_host + (size_t)r.src * fin, px, C);
            continue;
        }
        if (prepared != r.pair) {
            // frame p+1 of the previous pair is frame p of this one: keep it on the device
            int a = -1, b = -1;
            for (int k = 0; k < 2; ++k) {
                if (have[k] == r.pair) a = k;
                if (have[k] == r.pair + 1) b = k;
            }
            if (a < 0) {
                a = b == 0 ? 1 : 0;
                VFI_CHECK_HIP(hipMemcpyAsync(f[a].p, frames_host + (size_t)r.pair * fin, fin * sizeof(float), hipMemcpyHostToDevice, st));
                have[a] = r.pair;
            }
            if (b < 0) {
                b = a ^ 1;
                VFI_CHECK_HIP(hipMemcpyAsync(f[b].p, frames_host + (size_t)(r.pair + 1) * fin, fin * sizeof(float), hipMemcpyHostToDevice, st));
                have[b] = r.pair + 1;
            }
            if (int rc = vfi_m2m_prepare(net, f[a].p, f[b].p, C, H, W, st)) return rc;
            prepared = r.pair;
        }
        if (int rc = vfi_m2m_render(net, r.t, outd.p, st)) return rc;
        VFI_CHECK_HIP(hipMemcpyAsync(dst, outd.p, fout * sizeof(float), hipMemcpyDeviceToHost, st));
        VFI_CHECK_HIP(hipStreamSynchronize(st));
    }
    return 0;
}

}  // extern "C"
