// Internal helpers shared by the HIP translation units of libvfi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace vfi {

void set_error(const char* fmt, ...);
const char* get_error();

#define VFI_CHECK_HIP(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::vfi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                             __LINE__);                                                       \
            return -1;                                                                        \
        }                                                                                     \
    } while (0)

#define VFI_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            ::vfi::set_error(__VA_ARGS__); \
            return -2;                  \
        }                               \
    } while (0)

// ---- per-kernel event tracing ---------------------------------------------------------
bool trace_on();
void trace_begin(const char* name, hipStream_t s);
void trace_end(hipStream_t s);

struct TraceScope {
    hipStream_t s;
    bool on;
    TraceScope(const char* name, hipStream_t st) : s(st), on(trace_on()) {
        if (on) trace_begin(name, s);
    }
    ~TraceScope() {
        if (on) trace_end(s);
    }
};

// ---- shader-clock probe (vfi_clock_probe) -------------------------------------------------------------------------------------
// While a probe buffer is installed, every launch of the Winograd kernel takes one record (8 x u64, conv_wino.hip): workgroup 0 stamps
// s_memtime (shader cycles) and s_memrealtime (constant-rate counter) when it starts and when it ends.  The ratio of the two deltas is
// the clock the chip actually sustained while THIS kernel ran (it clocks to its power budget: 1.9-2.3 GHz under MFMA load,
// MI355X_MICROARCH.md "DVFS give-back").  clock_probe_tag: the host's index of the launch (its trace name is kept under it), -1 = off.
int clock_probe_tag(const char* name);
int clock_probe_install(unsigned long long* dev_records, int capacity);      // conv_wino.hip: the device-side state, current device
int wino_probe_read(unsigned* out32);                                         // conv_wino.hip: [4][8] stamp sums of the last probed launch

// ---- A/B options of tools/ and tests/ (include/vfi_hip_test.h: vfi_test_set_option) ----------------------------------------
// Every option selects between two CORRECT forms of a kernel or launch (fused / unfused, tile variant, ...).  They are set through
// the test C entry point only — the product library reads NO experiment switch from the environment, so a stray variable cannot
// change a frame (tests/test_gpu_env_hygiene.py).  Values are read at every launch (one relaxed atomic load).
enum Option {
    kOptStageQuad,       // bit mask of the next-block scales (2, 4, 8) whose transition takes the quad kernel (default 14 = all)
    kOptFuseEncode,      // 1: RIFE 4.7 frame pack in one launch (default), 0: prep + encode.0 + encode.1 as three kernels
    kOptFuse0a,          // 1: block 2->3 transition fused with block 3's conv0.0 (default), 0: separate
    kOptM2n2Px,          // pixel count from which wide direct-conv layers take the m2n2 tile (default -1 = never)
    kOptGroupedVariant,  // forced tile variant of the grouped (transposed) convolution (default -1 = heuristic)
    kOptSplitK,          // 1: split-K allowed for layer objects (default), 0: never
    kOptSplatAtomic,     // 1: force the LDS-atomic splat kernel, 3: staged list gather with compaction for C == 4 (default 0: list-gather kernel with the atomic kernel as overflow fallback)
    kOptSplatSpillCap,   // spill-list capacity of the list-gather splat (default -1 = built-in)
    kOptWinoXcd,         // 1: XCD-aware work order of the Winograd kernel (default), 0: plain order
    kOptDeconvWino,      // 1: ConvTranspose2d(4, 2, 1) + PixelShuffle (RIFE lastconv) as a 96-channel 3x3 layer on the Winograd kernel (default), 0: grouped direct kernel
    kOptEncodeBatched,   // 1: one frame-pack launch for a batch of frames where the caller offers one (default), 0: one launch per frame
    kOptWinoQuant,       // 1: layer objects leave launches of <= 2 rounds with a nearly empty last round to the direct kernel (default), 0: item count only
    kOptXcdBands,        // 1: XCD-aware workgroup order of the RIFE gather kernels (final blend, quad transitions), 0: plain order (default: r6 A/B measured the banded order 2-5 % SLOWER, profiles/r06_xcd_bands_ab.txt)
    kOptM2mFused,        // 1: M2M render as one kernel (m2m_render.hip; default), 0: splat inputs + summation splat + combine as separate launches
    kOptM2mSide,         // 1: M2M prepare runs the image-pyramid convolutions (EncDec's c features) on a side stream beside the PWC flow network, 0: one stream (default: r6 A/B measured the fork neutral on one pair — 7.00-7.04 vs 7.0-7.1 ms — and 8 % SLOWER under three pair lanes)
    kOptFilmSide,        // 1: FILM forward runs image 1's feature extraction and the backward flow pyramid on a side stream (another hardware queue) beside image 0's / the forward one, 0: one stream
    kOptWinoProbe,       // 1..4: the hot Winograd instantiation takes its cycle-ledger form (conv_wino.hip: g_wino_probe_out; default 0)
    kOptCount
};
long option(Option o);
// a new stream that the runtime has bound to another hardware queue than `st` (decided by measurement with vfi_stream_spin; util.hip)
hipStream_t stream_apart_from(hipStream_t st);
void stream_give_back(hipStream_t side);      // to the idle list stream_apart_from draws its candidates from (never destroyed)
int option_set(const char* name, long value);      // 0, or -2 for an unknown name
int variant_override(const char* trace_name);      // tile variant forced for a trace name (vfi_test_variant_override), -1 = none

// Compute units the persistent kernels may size their grids for: the device's count minus what vfi_set_reserved_cus holds back for a
// collective kernel running beside them (a resident RCCL kernel takes whole CUs from one-workgroup-per-CU kernels, whose displaced
// workgroups then run as a second round: +37 % while it is resident, profiles/r04_reserved_cus.txt).  A multiple of 8 (XCDs), >= 8.
int launch_cus(int device_cus);

// the 4-channel summation splat on the staged list-gather machinery of m2m_render.hip (round 6): in / out [N,H,W,4], flow [N,H,W,2]
int softsplat4_launch(const float* in, const float* flow, float* out, int N, int H, int W, hipStream_t s);

constexpr int kMaxDevices = 16;   // devices one process may drive (per-device caches are indexed by the HIP device id)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- convolution on the fp32 matrix cores -----------------------------------------------
// A convolution "group" is a set of taps sharing one weight block; a plain 3x3 conv has one
// group of 9 taps, the 4x4/stride-2 transposed conv has 4 groups (output parities) of 4 taps.
struct ConvArgs {
    const float* in;    // [N,Hin,Win,in_cs] (channels >= Cin_p readable, zero beyond Cin)
    const float* w;     // packed [group][tap][Cin_p/8][Cout_p][8]
    const float* bias;  // [group][Cout_p]
    const float* beta;  // [Cout_p] or nullptr       y = act((conv+bias)*beta + res)
    const float* res;   // [N,Hout,Wout,res_cs] or nullptr
    float* out;         // [N,Hout,Wout,out_cs]; group g writes channels [g*Cout_p, g*Cout_p+Cout)
    int N, Hin, Win, in_cs;
    int Hout, Wout, out_cs, res_cs;
    int Cin_p, Cout_p, Cout;
    int ntaps;
    int act;      // 0 none, 1 leaky relu(slope), 2 clamp to [0,1], 3 PReLU per channel (prelu[]), 4 sigmoid
    float slope;
    int tiles_x, tiles_y;  // output tiles per image (filled by the launcher)
    int tap_y0, tap_x0;    // origin of the tap rectangle (3x3: -1,-1), ignored for grouped
    int in_plane;          // 0: input is NHWC [.,.,in_cs]; >0: planar4 input, floats between 4-channel planes
                           //    (image = in_cs/4 planes of [Hin][Win][4])
    int out_mode;          // 0: NHWC store; 1 (grouped only): PixelShuffle(2) of the transposed conv, planar4
                           //    [N][2][4*Hout][4*Wout][4] (channel c of Cout/4 -> plane c/4, component c%4);
                           //    2 (grouped only): plain transposed-conv output, NHWC [N][2*Hout][2*Wout][out_cs]
    int out_planes;        // out_mode 1: planes per image of the planar4 output (0 = 2)
    int pad_replicate;     // 0: zero padding; 1: replicate (edge clamp) padding of the input
    const float* prelu;    // [Cout_p] per-channel negative slopes (act == 3)
    float post_scale, post_shift;  // y = act(...) * post_scale + post_shift  (post_scale == 0 means "not set" = 1, 0)
    int split_ok;          // caller: the launcher may cut K over several workgroups (split-K; generic layer objects only — the
                           //   RIFE network keeps one kernel per layer whatever the batch, so results do not depend on batching)
    const unsigned char* tapmask;  // 2x2 layers in the "up-sample x2 first" form (conv_mfma2.hip, MASKED): per 64-channel N block the taps (bit a * 2 + b) of
    int par_cout;                  //   its parity; par_cout = channels per parity group (4 groups: Cout_p == 4 * par_cout), output [2 Hout, 2 Wout] interleaved
    int ksplit;            // set by the launcher: > 1 = split-K launch, blockIdx.z owns a K range and the slice
    long split_stride;     //   out + blockIdx.z * split_stride (floats) of the partial-sum workspace
};

struct ConvVariant {
    const char* name;
    int stride, taps, mt, nt, wm, wn, ck, grouped;
};
// Variant ids: 0.. = first generation (conv_mfma.hip, weights in registers),
// kConv2Base.. = second generation (conv_mfma2.hip, both operands through LDS-DMA).
constexpr int kConv2Base = 32;
int conv_num_variants();
const ConvVariant& conv_variant(int i);
int conv2_num_variants();
const ConvVariant& conv2_variant(int i);
int conv2_launch(const ConvArgs& a, int idx, hipStream_t s, const char* trace_name);
const ConvVariant* conv_variant_lookup(int id);
// Launch; variant < 0 selects by heuristic.  grouped convs must use a grouped variant.
int conv_pick_variant(const ConvArgs& a, int stride, bool grouped);
int conv_launch(const ConvArgs& a, int stride, bool grouped, int variant, hipStream_t s,
                const char* trace_name);

// Host-side weight packing.  w_oihw: [Cout][Cin][3][3] -> packed (1 group, 9 taps).
void pack_conv3x3(const float* w_oihw, const float* bias, int Cout, int Cin, int Cin_p,
                  int Cout_p, std::vector<float>& wp, std::vector<float>& bp);
// w_iohw: [Cin][Cout][4][4] (ConvTranspose2d, stride 2, pad 1) -> 4 groups x 4 taps, Cout_p = 32k.
void pack_deconv4x4(const float* w_iohw, const float* bias, int Cin, int Cout, int Cin_p,
                    int Cout_p, std::vector<float>& wp, std::vector<float>& bp);
void conv3x3_taps(ConvArgs& a);

// ---- Winograd F(2x2,3x3) form of the 3x3 stride-1 convolution (conv_wino.hip) ------------------------------------------------
// Weights: U = G g G^T, packed [Cout_p/32][Cin_p/8][j][xi/4][half][co%32][xi%4] (16 * Cin_p * Cout_p floats); chan_map translates
// logical to physical input channels (nullptr = identity).  Launch: ConvArgs as for the direct kernel with a.w = that pack;
// variant 0 = pick the region shape, 8 / 16 = 16x8 / 32x4 output pixels per wave.
void pack_wino3x3(const float* w_oihw, int Cout, int Cin, const int* chan_map, int Cin_p, int Cout_p, std::vector<float>& wp);
void pack_deconv_as_conv3x3(const float* w_iohw, const float* bias, int Cin, int LO, std::vector<float>& w3, std::vector<float>& b3);
bool conv_wino_eligible(const ConvArgs& a);
int conv_wino_mode(int set);      // set < 0: query.  0 automatic, 1 direct kernel only, 2 Winograd wherever legal
int conv_wino_launch(const ConvArgs& a, int variant, hipStream_t s, const char* trace_name);
void deconv4x4_taps(ConvArgs& a);

int conv_naive_launch(const float* in, const float* w_oihw_dev, const float* bias_dev,
                      const float* beta_dev, float* out, int N, int H, int W, int Cin, int in_cs,
                      int Cout, int stride, int act, float slope, hipStream_t s);

}  // namespace vfi
