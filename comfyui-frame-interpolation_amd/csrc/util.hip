// Error string, device selection and per-kernel event tracing for libvfi_hip.so.
#include "vfi_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vfi_hip.h"
#include "../../include/vfi_hip_test.h"

namespace vfi {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

// ---- A/B options (vfi_common.h: enum Option) ---------------------------------------------------------------------------------
static const struct { const char* name; long dflt; } kOptTable[kOptCount] = {
    {"stage_quad", 14}, {"fuse_encode", 1}, {"fuse0a", 1}, {"m2n2_px", -1}, {"grouped_variant", -1}, {"splitk", 1},
    {"splat_atomic", 0}, {"splat_spill_cap", -1}, {"wino_xcd", 1}, {"deconv_wino", 1}, {"encode_batched", 1}, {"wino_quant", 1}, {"xcd_bands", 0}, {"m2m_fused", 1}, {"m2m_side", 0}, {"film_side", 1}, {"wino_probe", 0},
};
static std::atomic<long> g_opt[kOptCount];
static std::atomic<bool> g_opt_init{false};
static std::mutex g_opt_mu;
static std::map<std::string, int> g_variant_override;
static std::atomic<int> g_variant_override_n{0};
static void opt_init() {
    if (g_opt_init.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (g_opt_init.load(std::memory_order_relaxed)) return;
    for (int i = 0; i < kOptCount; ++i) g_opt[i].store(kOptTable[i].dflt, std::memory_order_relaxed);
    g_opt_init.store(true, std::memory_order_release);
}
long option(Option o) {
    opt_init();
    return g_opt[o].load(std::memory_order_relaxed);
}
#ifdef VFI_TEST_TAPS      // the product library has no way to leave the defaults (include/vfi_hip_test.h; csrc/build.py builds the test library)
int option_set(const char* name, long value) {
    opt_init();
    for (int i = 0; i < kOptCount; ++i)
        if (name && !strcmp(name, kOptTable[i].name)) {
            g_opt[i].store(value, std::memory_order_relaxed);
            return 0;
        }
    set_error("vfi_test_set_option: unknown option '%s'", name ? name : "(null)");
    return -2;
}
#endif
int variant_override(const char* trace_name) {
    if (!trace_name || !g_variant_override_n.load(std::memory_order_acquire)) return -1;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    auto it = g_variant_override.find(trace_name);
    return it == g_variant_override.end() ? -1 : it->second;
}
#ifdef VFI_TEST_TAPS
static void variant_override_set(const char* spec) {      // "conv0a_b3=42,resconv_c128=36"; empty / null clears
    std::lock_guard<std::mutex> lk(g_opt_mu);
    g_variant_override.clear();
    std::string str = spec ? spec : "";
    size_t pos = 0;
    while (pos < str.size()) {
        const size_t comma = str.find(',', pos), end = comma == std::string::npos ? str.size() : comma;
        const size_t eq = str.find('=', pos);
        if (eq != std::string::npos && eq < end) g_variant_override[str.substr(pos, eq - pos)] = atoi(str.c_str() + eq + 1);
        pos = end + 1;
    }
    g_variant_override_n.store((int)g_variant_override.size(), std::memory_order_release);
}
#endif

struct TraceRec {
    const char* name;
    hipEvent_t e0, e1;
};
static bool g_trace = false;
static std::vector<TraceRec> g_recs;
static std::vector<hipEvent_t> g_free_events;

static hipEvent_t get_event() {
    if (!g_free_events.empty()) {
        hipEvent_t e = g_free_events.back();
        g_free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

bool trace_on() { return g_trace; }
void trace_begin(const char* name, hipStream_t s) {
    TraceRec r;
    r.name = name;
    r.e0 = get_event();
    r.e1 = get_event();
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void trace_end(hipStream_t s) {
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().e1, s);
}


}  // namespace vfi

using namespace vfi;

namespace vfi {
// A new stream that the HIP runtime has bound to another hardware queue than `st`.  The runtime multiplexes streams onto 4 hardware queues,
// bound at first use; two streams of one queue run strictly in turn, so a fork onto such a stream overlaps nothing.  Decided by
// measurement: a spinning workgroup on each (vfi_stream_spin), 1x the spin apart, 2x together.  Rejected candidates stay alive (the next
// one is then bound elsewhere); after 8 tries the last one is used as it is (right frames, no overlap).
static std::mutex g_side_mu;
static std::vector<hipStream_t> g_side_idle;      // side streams nobody uses at the moment (rejected candidates, returned ones): never destroyed

static bool streams_together(hipStream_t st, hipStream_t c) {
    vfi_stream_spin(c, 1);
    vfi_stream_spin(st, 1);
    int together = 0;
    for (int r = 0; r < 3; ++r) {
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(c);
        const auto t0 = std::chrono::steady_clock::now();
        vfi_stream_spin(st, 300);
        vfi_stream_spin(c, 300);
        (void)hipStreamSynchronize(st);
        (void)hipStreamSynchronize(c);
        together += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 480.0;
    }
    return together >= 2;
}

hipStream_t stream_apart_from(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::vector<hipStream_t> rejected;
    hipStream_t found = nullptr;
    for (int t = 0; t < 12 && !found; ++t) {
        hipStream_t c = nullptr;
        // idle streams of this device first (an object that is rebuilt per call must not create streams per call)
        for (size_t i = 0; i < g_side_idle.size() && !c; ++i) {
            int d = -1;
            if (hipStreamGetDevice(g_side_idle[i], &d) == hipSuccess && d == dev) {
                c = g_side_idle[i];
                g_side_idle.erase(g_side_idle.begin() + i);
            }
        }
        if (!c && hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) break;
        if (streams_together(st, c)) rejected.push_back(c);
        else found = c;
    }
    if (!found && !rejected.empty()) {      // right frames, no overlap
        found = rejected.back();
        rejected.pop_back();
    }
    for (hipStream_t r : rejected) g_side_idle.push_back(r);
    return found;
}

void stream_give_back(hipStream_t side) {
    if (!side) return;
    (void)hipStreamSynchronize(side);
    std::lock_guard<std::mutex> lk(g_side_mu);
    g_side_idle.push_back(side);
}

// spins until `ticks` of s_memrealtime (100 MHz on gfx950) have passed: vfi_stream_spin
__global__ void stream_spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace vfi

extern "C" {

int vfi_init(int device) {
    int n = 0;
    VFI_CHECK_HIP(hipGetDeviceCount(&n));
    VFI_REQUIRE(device >= 0 && device < n, "vfi_init: device %d out of range (%d visible)", device, n);
    VFI_CHECK_HIP(hipSetDevice(device));
    return 0;
}

const char* vfi_last_error(void) { return get_error(); }

int vfi_device_info(char* arch_buf, int arch_buf_len, int* n_cus) {
    int dev = 0;
    VFI_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    VFI_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (arch_buf && arch_buf_len > 0) {
        strncpy(arch_buf, p.gcnArchName, arch_buf_len - 1);
        arch_buf[arch_buf_len - 1] = 0;
    }
    if (n_cus) *n_cus = p.multiProcessorCount;
    return 0;
}

static std::atomic<int> g_reserved_cus{0};
}  // extern "C" (reopened below)
namespace vfi {
int launch_cus(int device_cus) {
    const int r = g_reserved_cus.load(std::memory_order_relaxed);
    if (r <= 0) return device_cus;      // nothing reserved: the device's own count, whatever its divisibility
    int c = device_cus - r;
    c -= c % 8;
    return c < 8 ? (device_cus < 8 ? device_cus : 8) : c;
}
}  // namespace vfi
extern "C" {

int vfi_set_reserved_cus(int n) {
    VFI_REQUIRE(n >= 0 && n <= 1024, "vfi_set_reserved_cus: %d compute units", n);
    g_reserved_cus.store(n, std::memory_order_relaxed);
    return 0;
}

int vfi_get_reserved_cus(void) { return g_reserved_cus.load(std::memory_order_relaxed); }

int vfi_stream_create(void** stream_out) {
    VFI_REQUIRE(stream_out, "vfi_stream_create: null argument");
    hipStream_t s = nullptr;
    VFI_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (void*)s;
    return 0;
}

int vfi_stream_destroy(void* stream) {
    VFI_REQUIRE(stream, "vfi_stream_destroy: null stream");
    VFI_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

int vfi_stream_spin(void* stream, int microseconds) {
    VFI_REQUIRE(microseconds > 0 && microseconds <= 100000, "vfi_stream_spin: %d us out of range (1 .. 100000)", microseconds);
    // launched with the stream's device current (a host thread that never chose a device probes streams of any device); restored afterwards
    int cur = -1, sdev_i = -1;
    if (stream) {
        hipDevice_t sdev = 0;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && hipGetDevice(&cur) == hipSuccess && cur != (int)sdev) {
            sdev_i = (int)sdev;
            VFI_CHECK_HIP(hipSetDevice(sdev_i));
        } else {
            (void)hipGetLastError();
        }
    }
    hipLaunchKernelGGL(vfi::stream_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull);
    const hipError_t le = hipGetLastError();
    if (sdev_i >= 0 && cur >= 0) (void)hipSetDevice(cur);
    VFI_CHECK_HIP(le);
    return 0;
}

int vfi_memcpy_async(void* dst, const void* src, int64_t bytes, int kind, void* stream) {
    VFI_REQUIRE(dst && src && bytes >= 0 && (kind == 1 || kind == 2 || kind == 3), "vfi_memcpy_async: bad arguments (kind %d)", kind);
    if (bytes == 0) return 0;
    // worker threads never chose a device: the copy must be issued with the stream's device current.  The caller's current device is
    // restored afterwards (a public entry point must not leave a side effect on the calling thread — ADVICE r4; comm.hip's DeviceScope)
    int cur = -1, sdev_i = -1;
    if (stream) {
        hipDevice_t sdev = 0;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) != hipSuccess || hipGetDevice(&cur) != hipSuccess) {
            (void)hipGetLastError();      // (only these queries' own failure is read away: the copy below then reports what is wrong with the stream)
            cur = -1;
        } else if (cur != (int)sdev) {
            sdev_i = (int)sdev;
            VFI_CHECK_HIP(hipSetDevice(sdev_i));
        }
    }
    const hipError_t ce = hipMemcpyAsync(dst, src, (size_t)bytes, kind == 1 ? hipMemcpyHostToDevice : (kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice),
                                         (hipStream_t)stream);
    if (sdev_i >= 0 && cur >= 0) (void)hipSetDevice(cur);
    VFI_CHECK_HIP(ce);
    return 0;
}

#ifdef VFI_TEST_TAPS
int vfi_test_set_option(const char* name, int64_t value) { return option_set(name, (long)value); }
int vfi_test_variant_override(const char* spec) {
    variant_override_set(spec);
    return 0;
}
#endif

int vfi_trace_enable(int on) {
    g_trace = on != 0;
    return 0;
}

int vfi_trace_reset(void) {
    for (auto& r : g_recs) {
        g_free_events.push_back(r.e0);
        g_free_events.push_back(r.e1);
    }
    g_recs.clear();
    return 0;
}

int vfi_trace_report(char* buf, int buf_len) {
    std::map<std::string, std::pair<int, double>> agg;
    std::vector<std::string> order;
    for (auto& r : g_recs) {
        VFI_CHECK_HIP(hipEventSynchronize(r.e1));
        float ms = 0.f;
        VFI_CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        auto it = agg.find(r.name);
        if (it == agg.end()) {
            agg[r.name] = {1, (double)ms};
            order.push_back(r.name);
        } else {
            it->second.first += 1;
            it->second.second += ms;
        }
    }
    std::string out;
    char line[256];
    for (auto& k : order) {
        snprintf(line, sizeof(line), "%s %d %.6f\n", k.c_str(), agg[k].first, agg[k].second);
        out += line;
    }
    VFI_REQUIRE((int)out.size() + 1 <= buf_len, "vfi_trace_report: buffer too small (%d needed)", (int)out.size() + 1);
    memcpy(buf, out.c_str(), out.size() + 1);
    return 0;
}

}  // extern "C"
