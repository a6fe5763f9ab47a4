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

struct vfi_comm {
    int n = 0;
    int devices[kMaxDevices];
    ncclComm_t comms[kMaxDevices];
};

extern "C" {

vfi_comm_t* vfi_comm_create(int n_devices, const int* devices) {
    if (n_devices < 1 || n_devices > kMaxDevices || !devices) {
        set_error("vfi_comm_create: %d devices (1..%d)", n_devices, kMaxDevices);
        return nullptr;
    }
    const RcclApi& api = rccl();
    if (!api.ok) {
        set_error("vfi_comm_create: RCCL not available (dlopen librccl.so.1: %s)", api.handle ? "missing symbols" : dlerror());
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
    delete c;
}

int vfi_comm_size(const vfi_comm_t* c) { return c ? c->n : 0; }

int vfi_comm_broadcast(vfi_comm_t* c, float* const* bufs_dev, int64_t count, int root, void* const* streams) {
    VFI_REQUIRE(c && bufs_dev && streams && count >= 0 && root >= 0 && root < c->n, "vfi_comm_broadcast: bad arguments");
    if (count == 0) return 0;     // (a one-device clique still goes through RCCL: the degenerate case exercises the same calls)
    const RcclApi& api = rccl();
    VFI_CHECK_NCCL(api.GroupStart());
    for (int i = 0; i < c->n; ++i)
        VFI_CHECK_NCCL(api.Broadcast(bufs_dev[root], bufs_dev[i], (size_t)count, ncclFloat32, root, c->comms[i], (hipStream_t)streams[i]));
    VFI_CHECK_NCCL(api.GroupEnd());
    return 0;
}

int vfi_comm_all_gather_v(vfi_comm_t* c, float* const* bufs_dev, const int64_t* counts, void* const* streams) {
    VFI_REQUIRE(c && bufs_dev && counts && streams, "vfi_comm_all_gather_v: bad arguments");
    const RcclApi& api = rccl();
    // one group: for every root r, its block [prefix[r], prefix[r] + counts[r]) travels to the same place of every buffer
    VFI_CHECK_NCCL(api.GroupStart());
    int64_t off = 0;
    for (int r = 0; r < c->n; ++r) {
        VFI_REQUIRE(counts[r] >= 0, "vfi_comm_all_gather_v: negative count");
        if (counts[r] > 0)
            for (int i = 0; i < c->n; ++i)
                VFI_CHECK_NCCL(api.Broadcast(bufs_dev[r] + off, bufs_dev[i] + off, (size_t)counts[r], ncclFloat32, r, c->comms[i],
                                             (hipStream_t)streams[i]));
        off += counts[r];
    }
    VFI_CHECK_NCCL(api.GroupEnd());
    return 0;
}

}  // extern "C"
