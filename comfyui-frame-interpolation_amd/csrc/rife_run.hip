// vfi_rife_run — the whole RIFE node call for a HOST clip behind one C entry point: scheduling, uploads,
// the per-task hot loop and the output interleave, for host applications that do not want to re-implement the node loop
// (vfi_models/rife/__init__.py:149-239) on top of vfi_rife_load_frame / vfi_rife_interpolate.  Built ONLY on the public C
// ABI of include/vfi_hip.h plus the HIP runtime.
//
//   schedule  : per-pair multipliers (m <= 1 or a skipped pair: the frame is kept, no new frames; rife/__init__.py:164-174),
//               tasks (pair, k/m) in pair order, batches of `batch` tasks;
//   output    : frame_0, its new frames, frame_1, ..., frame_last (rife/__init__.py:225-230), alpha dropped, new frames
//               clamped to [0,1] by the network's last kernel;
//   pipeline  : pinned staging both ways, three streams (upload / compute / download); every input frame is uploaded and
//               encoded once; while batch i computes, the host copies batch i-1's frames into their output rows.
// Single host thread: simple and correct, not the fastest host path (the Python node's hostpipe.py runs the staging copies
// on worker threads).
#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/vfi_hip.h"
#include "vfi_common.h"

using namespace vfi;

namespace {

struct Task {
    int pair;
    float t;
    int64_t row;   // output row of the new frame
};

struct RunWs {   // per-device staging, grown on demand, kept for the life of the process
    hipStream_t st = nullptr, su = nullptr, sd = nullptr;
    std::vector<float*> up_host, up_dev;
    std::vector<hipEvent_t> up_ready, up_consumed;
    size_t up_floats = 0;
    float* out_dev[2] = {nullptr, nullptr};
    float* out_host[2] = {nullptr, nullptr};
    hipEvent_t comp[2] = {nullptr, nullptr}, down[2] = {nullptr, nullptr};
    size_t out_floats = 0;
};
RunWs g_ws[kMaxDevices];

int ensure_ws(RunWs& w, int n_up, size_t up_floats, size_t out_floats) {
    if (!w.st) {
        VFI_CHECK_HIP(hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking));
        VFI_CHECK_HIP(hipStreamCreateWithFlags(&w.su, hipStreamNonBlocking));
        VFI_CHECK_HIP(hipStreamCreateWithFlags(&w.sd, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            VFI_CHECK_HIP(hipEventCreateWithFlags(&w.comp[k], hipEventDisableTiming));
            VFI_CHECK_HIP(hipEventCreateWithFlags(&w.down[k], hipEventDisableTiming));
        }
    }
    if (w.up_floats < up_floats) {   // frame size grew: drop the ring
        for (float* p : w.up_host) (void)hipHostFree(p);
        for (float* p : w.up_dev) (void)hipFree(p);
        w.up_host.clear();
        w.up_dev.clear();
        w.up_floats = up_floats;
    }
    while ((int)w.up_host.size() < n_up) {
        float *h = nullptr, *d = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        VFI_CHECK_HIP(hipHostMalloc((void**)&h, w.up_floats * sizeof(float), hipHostMallocDefault));
        VFI_CHECK_HIP(hipMalloc((void**)&d, w.up_floats * sizeof(float)));
        w.up_host.push_back(h);
        w.up_dev.push_back(d);
        if (w.up_ready.size() < w.up_host.size()) {
            VFI_CHECK_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
            VFI_CHECK_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            w.up_ready.push_back(e0);
            w.up_consumed.push_back(e1);
        }
    }
    if (w.out_floats < out_floats) {
        for (int k = 0; k < 2; ++k) {
            if (w.out_dev[k]) (void)hipFree(w.out_dev[k]);
            if (w.out_host[k]) (void)hipHostFree(w.out_host[k]);
            VFI_CHECK_HIP(hipMalloc((void**)&w.out_dev[k], out_floats * sizeof(float)));
            VFI_CHECK_HIP(hipHostMalloc((void**)&w.out_host[k], out_floats * sizeof(float), hipHostMallocDefault));
        }
        w.out_floats = out_floats;
    }
    return 0;
}

// out_row[H*W*3] = frame[H*W*C] without alpha
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

}  // namespace

extern "C" int vfi_rife_run(vfi_rife_t* net, const float* frames_host, int N, int H, int W, int C, const int* multipliers,
                            const uint8_t* skip, float scale_factor, int batch, float* out_host, int64_t* n_out) {
    VFI_REQUIRE(net && n_out && N >= 1 && H > 0 && W > 0 && C >= 3, "vfi_rife_run: bad arguments (N=%d H=%d W=%d C=%d)", N, H, W, C);
    VFI_REQUIRE(batch >= 1 && batch <= 32, "vfi_rife_run: batch %d outside 1..32", batch);
    // ---- schedule (rife/__init__.py:149-174) and output rows (:225-230)
    std::vector<Task> tasks;
    std::vector<int64_t> src_row(N);
    int64_t row = 0;
    for (int p = 0; p < N; ++p) {
        src_row[p] = row++;
        if (p == N - 1) break;
        const int m = multipliers ? multipliers[p] : 2;
        if ((skip && skip[p]) || m <= 1) continue;
        for (int k = 1; k < m; ++k) tasks.push_back({p, (float)k / (float)m, row++});
    }
    *n_out = row;
    if (!out_host) return 0;   // size query
    VFI_REQUIRE(frames_host, "vfi_rife_run: null frames");
    const size_t px = (size_t)H * W, fin = px * C, fout = px * 3;
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    VFI_REQUIRE(dev >= 0 && dev < kMaxDevices, "vfi_rife_run: device index %d out of range", dev);
    RunWs& w = g_ws[dev];
    const int n_slots = 2 * batch + 2, n_up = batch + 2;
    if (!tasks.empty()) {
        if (int rc = vfi_rife_configure(net, H, W, batch, n_slots, scale_factor)) return rc;
        if (int rc = ensure_ws(w, n_up, fin, (size_t)batch * fout)) return rc;
    }
    // frame -> network slot, evicting frames the current batch does not need (tasks ascend by pair)
    std::map<int, int> slot_of;
    std::vector<int> free_slots;
    for (int s = n_slots - 1; s >= 0; --s) free_slots.push_back(s);
    int up_count = 0;
    std::vector<char> up_used(n_up, 0);
    struct Pending {
        size_t pos, n;
        bool live = false;
    } pend[2];
    auto drain = [&](int k) -> int {   // batch in buffer k: wait for its D2H, move the frames to their rows
        if (!pend[k].live) return 0;
        VFI_CHECK_HIP(hipEventSynchronize(w.down[k]));
        for (size_t i = 0; i < pend[k].n; ++i)
            memcpy(out_host + (size_t)tasks[pend[k].pos + i].row * fout, w.out_host[k] + i * fout, fout * sizeof(float));
        pend[k].live = false;
        return 0;
    };
    int next_src = 0;   // pass-through frames are copied in between, up to the pair the pipeline has reached
    auto copy_src_until = [&](int last) {
        for (; next_src <= last && next_src < N; ++next_src)
            copy_rgb(out_host + (size_t)src_row[next_src] * fout, frames_host + (size_t)next_src * fin, px, C);
    };
    int k = 0;
    for (size_t pos = 0; pos < tasks.size(); pos += batch, k ^= 1) {
        const size_t nb = std::min((size_t)batch, tasks.size() - pos);
        if (int rc = drain(k)) return rc;   // buffer k was used two batches ago
        std::vector<int> need;
        for (size_t i = 0; i < nb; ++i)
            for (int f : {tasks[pos + i].pair, tasks[pos + i].pair + 1})
                if (std::find(need.begin(), need.end(), f) == need.end()) need.push_back(f);
        for (auto it = slot_of.begin(); it != slot_of.end();) {
            if (std::find(need.begin(), need.end(), it->first) == need.end()) {
                free_slots.push_back(it->second);
                it = slot_of.erase(it);
            } else {
                ++it;
            }
        }
        for (int f : need) {
            if (slot_of.count(f)) continue;
            VFI_REQUIRE(!free_slots.empty(), "vfi_rife_run: frame cache exhausted");
            const int slot = free_slots.back();
            free_slots.pop_back();
            slot_of[f] = slot;
            const int r = up_count++ % n_up;
            if (up_used[r]) VFI_CHECK_HIP(hipEventSynchronize(w.up_consumed[r]));   // the ring slot's previous frame is encoded
            memcpy(w.up_host[r], frames_host + (size_t)f * fin, fin * sizeof(float));
            VFI_CHECK_HIP(hipMemcpyAsync(w.up_dev[r], w.up_host[r], fin * sizeof(float), hipMemcpyHostToDevice, w.su));
            VFI_CHECK_HIP(hipEventRecord(w.up_ready[r], w.su));
            VFI_CHECK_HIP(hipStreamWaitEvent(w.st, w.up_ready[r], 0));
            if (int rc = vfi_rife_load_frame(net, slot, w.up_dev[r], C, w.st)) return rc;
            VFI_CHECK_HIP(hipEventRecord(w.up_consumed[r], w.st));
            up_used[r] = 1;
        }
        int s0[32], s1[32];
        float ts[32];
        for (size_t i = 0; i < nb; ++i) {
            s0[i] = slot_of[tasks[pos + i].pair];
            s1[i] = slot_of[tasks[pos + i].pair + 1];
            ts[i] = tasks[pos + i].t;
        }
        if (int rc = vfi_rife_interpolate(net, (int)nb, s0, s1, ts, w.out_dev[k], w.st)) return rc;
        VFI_CHECK_HIP(hipEventRecord(w.comp[k], w.st));
        VFI_CHECK_HIP(hipStreamWaitEvent(w.sd, w.comp[k], 0));
        VFI_CHECK_HIP(hipMemcpyAsync(w.out_host[k], w.out_dev[k], nb * fout * sizeof(float), hipMemcpyDeviceToHost, w.sd));
        VFI_CHECK_HIP(hipEventRecord(w.down[k], w.sd));
        pend[k].pos = pos, pend[k].n = nb, pend[k].live = true;
        // host work under the GPU's: the previous batch's frames, then the pass-through frames up to this batch's last pair
        if (int rc = drain(k ^ 1)) return rc;
        copy_src_until(tasks[pos + nb - 1].pair);
    }
    if (int rc = drain(0)) return rc;
    if (int rc = drain(1)) return rc;
    copy_src_until(N - 1);
    if (!tasks.empty()) VFI_CHECK_HIP(hipStreamSynchronize(w.st));
    return 0;
}
