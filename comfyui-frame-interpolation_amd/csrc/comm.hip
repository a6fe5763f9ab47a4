// Single-process multi-device communicator over RCCL / xGMI: one process (ComfyUI's prompt worker) drives
// every visible GPU, one host thread and one stream per device; ncclCommInitAll builds the clique.
//
//   * weights: broadcast ONCE from the device that packed them, as one flat buffer (21.3 MB for RIFE 4.7) —
//     vfi_comm_broadcast over the weight arenas of vfi_rife_weights();
//   * new frames: all-gather-v, in place, as grouped per-root ncclBroadcasts (rank contributions may differ: P mod R, skip
//     lists, per-pair multipliers) — vfi_comm_all_gather_v.  The node itself does not need it: each device copies its own
//     shard straight into the shared host output tensor over its own PCIe link (multidev.py); it is there for device-side
//     consumers and for bench.py's throughput definition.
//
// RCCL is bound at first use (dlopen of librccl.so.1 and its entry points): the single-GPU product path neither links nor
// loads it.  All calls for all devices are issued by ONE host thread inside ncclGroupStart / ncclGroupEnd, on the streams the
// caller passes (one per device, created on that device).
#include "vfi_common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include <vector>

#include "../../include/vfi_hip.h"

namespace vfi {

struct RcclApi {
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    void* handle = nullptr;
    bool ok = false;
};

static RcclApi g_rccl;
static std::once_flag g_rccl_once;

static const RcclApi& rccl() {
    std::call_once(g_rccl_once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
        if (!g_rccl.handle) return;
        auto sym = [&](const char* n) { return dlsym(g_rccl.handle, n); };
        g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
        g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
        g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
        g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
        g_rccl.ok = g_rccl.CommInitAll && g_rccl.CommDestroy && g_rccl.Broadcast && g_rccl.GroupStart && g_rccl.GroupEnd &&
                    g_rccl.GetErrorString;
    });
    return g_rccl;
}

#define VFI_CHECK_NCCL(expr)                                                                                        \
    do {                                                                                                            \
        ncclResult_t _r = (expr);                                                                                   \
        if (_r != ncclSuccess) {                                                                                    \
            ::vfi::set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__);          \
            return -1;                                                                                              \
        }                                                                                                           \
    } while (0)

}  // namespace vfi

using namespace vfi;

// Restores the calling thread's current device on every way out (the loops below hop over the clique's devices; an early error
// return must not leave later torch / HIP work of this thread on some rank's GPU).
struct DeviceScope {
    int prev = 0;
    DeviceScope() { (void)hipGetDevice(&prev); }
    ~DeviceScope() { (void)hipSetDevice(prev); }
};

struct vfi_comm {
    int n = 0;
    int devices[kMaxDevices];
    ncclComm_t comms[kMaxDevices];
    // direct (full-mesh) all-gather: one copy stream and two events per ordered device pair, made on first use
    hipStream_t cs[kMaxDevices][kMaxDevices] = {};
    hipEvent_t ev_src[kMaxDevices] = {}, ev_dst[kMaxDevices] = {}, ev_done[kMaxDevices][kMaxDevices] = {};
    bool mesh_ready = false;
};

// An error between ncclGroupStart and ncclGroupEnd must still close the group: an open group queues every later collective of
// this thread and nothing launches any more (the next synchronize hangs).
#define VFI_NCCL_IN_GROUP(api, expr)                                                                              \
    do {                                                                                                            \
        ncclResult_t _r = (expr);                                                                                   \
        if (_r != ncclSuccess) {                                                                                    \
            (void)(api).GroupEnd();                                                                                 \
            ::vfi::set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString(_r), __FILE__, __LINE__);          \
            return -1;                                                                                              \
        }                                                                                                           \
    } while (0)

extern "C" {

vfi_comm_t* vfi_comm_create(int n_devices, const int* devices) {
    if (n_devices < 1 || n_devices > kMaxDevices || !devices) {
        set_error("vfi_comm_create: %d devices (1..%d)", n_devices, kMaxDevices);
        return nullptr;
    }
    const RcclApi& api = rccl();
    if (!api.ok) {
        const char* why = api.handle ? "missing symbols" : dlerror();
        set_error("vfi_comm_create: RCCL not available (dlopen librccl.so.1: %s)", why ? why : "unknown error");
        return nullptr;
    }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) visible = 0;
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= visible) {
            set_error("vfi_comm_create: device %d not visible (%d devices)", devices[i], visible);
            return nullptr;
        }
        for (int j = 0; j < i; ++j)
            if (devices[j] == devices[i]) {
                set_error("vfi_comm_create: device %d listed twice", devices[i]);
                return nullptr;
            }
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    vfi_comm* c = new vfi_comm();
    c->n = n_devices;
    for (int i = 0; i < n_devices; ++i) c->devices[i] = devices[i];
    const ncclResult_t r = api.CommInitAll(c->comms, n_devices, c->devices);
    (void)hipSetDevice(prev);
    if (r != ncclSuccess) {
        set_error("vfi_comm_create: ncclCommInitAll over %d devices failed: %s", n_devices, api.GetErrorString(r));
        delete c;
        return nullptr;
    }
    return c;
}

void vfi_comm_destroy(vfi_comm_t* c) {
    if (!c) return;
    for (int i = 0; i < c->n; ++i) (void)rccl().CommDestroy(c->comms[i]);
    {   // always walk the arrays: a mesh_init that failed half way leaves what it had created
        DeviceScope ds;
        for (int r = 0; r < c->n; ++r) {
            (void)hipSetDevice(c->devices[r]);
            if (c->ev_src[r]) (void)hipEventDestroy(c->ev_src[r]);
            if (c->ev_dst[r]) (void)hipEventDestroy(c->ev_dst[r]);
            for (int i = 0; i < c->n; ++i) {
                if (c->cs[r][i]) (void)hipStreamDestroy(c->cs[r][i]);
                if (c->ev_done[r][i]) (void)hipEventDestroy(c->ev_done[r][i]);
            }
        }
    }
    delete c;
}

int vfi_comm_size(const vfi_comm_t* c) { return c ? c->n : 0; }

int vfi_comm_broadcast(vfi_comm_t* c, float* const* bufs_dev, int64_t count, int root, void* const* streams) {
    VFI_REQUIRE(c && bufs_dev && streams && count >= 0 && root >= 0 && root < c->n, "vfi_comm_broadcast: bad arguments");
    if (count == 0) return 0;     // (a one-device clique still goes through RCCL: the degenerate case exercises the same calls)
    const RcclApi& api = rccl();
    VFI_CHECK_NCCL(api.GroupStart());
    for (int i = 0; i < c->n; ++i)
        VFI_NCCL_IN_GROUP(api, api.Broadcast(bufs_dev[root], bufs_dev[i], (size_t)count, ncclFloat32, root, c->comms[i], (hipStream_t)streams[i]));
    VFI_CHECK_NCCL(api.GroupEnd());
    return 0;
}

// The copies of an in-place all-gather with per-rank counts: rank r's block [prefix[r], prefix[r] + counts[r]) goes to the same
// place of every OTHER rank's buffer.  Pure function (no device): the direct path below executes this list, the CPU suite checks it.
int64_t vfi_comm_plan_all_gather(int n, const int64_t* counts, int64_t* plan, int64_t cap) {
    if (n < 1 || !counts || (!plan && cap > 0)) {
        set_error("vfi_comm_plan_all_gather: bad arguments");
        return -1;
    }
    int64_t off = 0, k = 0;
    for (int r = 0; r < n; ++r) {
        if (counts[r] < 0) {
            set_error("vfi_comm_plan_all_gather: negative count for rank %d", r);
            return -1;
        }
        if (counts[r] > 0)
            for (int i = 0; i < n; ++i) {
                if (i == r) continue;
                if (plan) {
                    if (4 * (k + 1) > cap) {
                        set_error("vfi_comm_plan_all_gather: plan buffer too small");
                        return -1;
                    }
                    plan[4 * k] = r, plan[4 * k + 1] = i, plan[4 * k + 2] = off, plan[4 * k + 3] = counts[r];
                }
                ++k;
            }
        off += counts[r];
    }
    return k;
}

static int mesh_init(vfi_comm* c) {
    if (c->mesh_ready) return 0;
    DeviceScope ds;
    // (a retry after a partial failure re-uses what already exists: every handle is created only while it is null)
    for (int r = 0; r < c->n; ++r) {
        VFI_CHECK_HIP(hipSetDevice(c->devices[r]));
        if (!c->ev_src[r]) VFI_CHECK_HIP(hipEventCreateWithFlags(&c->ev_src[r], hipEventDisableTiming));
        if (!c->ev_dst[r]) VFI_CHECK_HIP(hipEventCreateWithFlags(&c->ev_dst[r], hipEventDisableTiming));
        for (int i = 0; i < c->n; ++i) {
            if (i == r) continue;
            if (!c->cs[r][i]) VFI_CHECK_HIP(hipStreamCreateWithFlags(&c->cs[r][i], hipStreamNonBlocking));
            if (!c->ev_done[r][i]) VFI_CHECK_HIP(hipEventCreateWithFlags(&c->ev_done[r][i], hipEventDisableTiming));
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, c->devices[r], c->devices[i]) == hipSuccess && can) {
                // Any non-success return — hipErrorPeerAccessAlreadyEnabled included, the LIKELY one since RCCL's own same-process
                // P2P setup has usually enabled it — stays in the thread's sticky error slot until it is read: the next
                // VFI_CHECK_HIP(hipGetLastError()) behind a kernel launch would report it.  Read it away.  (A real failure only
                // means the copies stage through the host.)
                if (hipDeviceEnablePeerAccess(c->devices[i], 0) != hipSuccess) (void)hipGetLastError();
            } else {
                (void)hipGetLastError();
            }
        }
    }
    c->mesh_ready = true;
    return 0;
}

// Direct full-mesh form: xGMI is point to point, every pair of the node's 8 devices has its own link, so
// each rank pushes its block to its 7 peers with 7 concurrent peer copies (one stream per ordered pair): shard / link rate instead
// of a ring's (R-1) hops of shard / link rate.  Ordering: a copy r -> i starts after streams[r] (the block is computed) and after
// streams[i] (the destination's earlier readers of that slot are done); streams[i] then waits for all its incoming copies.
static int all_gather_direct(vfi_comm* c, float* const* bufs_dev, const int64_t* counts, void* const* streams) {
    if (mesh_init(c)) return -1;
    DeviceScope ds;
    for (int r = 0; r < c->n; ++r) {
        VFI_CHECK_HIP(hipSetDevice(c->devices[r]));
        VFI_CHECK_HIP(hipEventRecord(c->ev_src[r], (hipStream_t)streams[r]));
        VFI_CHECK_HIP(hipEventRecord(c->ev_dst[r], (hipStream_t)streams[r]));
    }
    std::vector<int64_t> plan((size_t)4 * c->n * c->n);
    const int64_t np = vfi_comm_plan_all_gather(c->n, counts, plan.data(), (int64_t)plan.size());
    if (np < 0) return -1;
    for (int64_t k = 0; k < np; ++k) {
        const int r = (int)plan[4 * k], i = (int)plan[4 * k + 1];
        const int64_t off = plan[4 * k + 2], cnt = plan[4 * k + 3];
        VFI_CHECK_HIP(hipSetDevice(c->devices[r]));
        VFI_CHECK_HIP(hipStreamWaitEvent(c->cs[r][i], c->ev_src[r], 0));
        VFI_CHECK_HIP(hipStreamWaitEvent(c->cs[r][i], c->ev_dst[i], 0));
        VFI_CHECK_HIP(hipMemcpyPeerAsync(bufs_dev[i] + off, c->devices[i], bufs_dev[r] + off, c->devices[r], (size_t)cnt * sizeof(float), c->cs[r][i]));
        VFI_CHECK_HIP(hipEventRecord(c->ev_done[r][i], c->cs[r][i]));
    }
    for (int64_t k = 0; k < np; ++k) {
        const int r = (int)plan[4 * k], i = (int)plan[4 * k + 1];
        VFI_CHECK_HIP(hipSetDevice(c->devices[i]));
        VFI_CHECK_HIP(hipStreamWaitEvent((hipStream_t)streams[i], c->ev_done[r][i], 0));
        // the source must not overwrite its block before the copy has read it
        VFI_CHECK_HIP(hipSetDevice(c->devices[r]));
        VFI_CHECK_HIP(hipStreamWaitEvent((hipStream_t)streams[r], c->ev_done[r][i], 0));
    }
    return 0;
}

// VFI_ALLGATHER = rccl (default: grouped per-root ncclBroadcast) | direct (per-link peer copies).  The direct form has only ever run
// on a clique of one device (no multi-GPU box has been available to this project): it stays opt-in until a 2+ device run has
// passed tests/test_gpu_multidev.py::test_direct_all_gather_then_kernel.  One of the supported runtime variables.
static int all_gather_mode() {
    static const int mode = [] {
        const char* e = getenv("VFI_ALLGATHER");
        return (e && (e[0] == 'd' || e[0] == 'D')) ? 0 : 1;
    }();
    return mode;
}

/* 0 = direct peer copies, 1 = RCCL grouped broadcasts: what vfi_comm_all_gather_v will use (bench.py reports it) */
int vfi_comm_all_gather_mode(void) { return all_gather_mode(); }

int vfi_comm_all_gather_v(vfi_comm_t* c, float* const* bufs_dev, const int64_t* counts, void* const* streams) {
    VFI_REQUIRE(c && bufs_dev && counts && streams, "vfi_comm_all_gather_v: bad arguments");
    for (int r = 0; r < c->n; ++r) VFI_REQUIRE(counts[r] >= 0, "vfi_comm_all_gather_v: negative count for rank %d", r);   // before any group opens
    if (all_gather_mode() == 0) return all_gather_direct(c, bufs_dev, counts, streams);
    const RcclApi& api = rccl();
    // one group: for every root r, its block [prefix[r], prefix[r] + counts[r]) travels to the same place of every buffer
    VFI_CHECK_NCCL(api.GroupStart());
    int64_t off = 0;
    for (int r = 0; r < c->n; ++r) {
        if (counts[r] > 0)
            for (int i = 0; i < c->n; ++i)
                VFI_NCCL_IN_GROUP(api, api.Broadcast(bufs_dev[r] + off, bufs_dev[i] + off, (size_t)counts[r], ncclFloat32, r, c->comms[i],
                                                     (hipStream_t)streams[i]));
        off += counts[r];
    }
    VFI_CHECK_NCCL(api.GroupEnd());
    return 0;
}

}  // extern "C"
