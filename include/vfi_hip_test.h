/*
 * vfi_hip_test.h — TEST TAPS: entry points that exist for the parity tests and A/B tools only (a naive cross-check convolution,
 * read-back of internal tensors, switches between two correct kernel forms).  They are NOT in the product library: csrc/build.py
 * compiles them (`#ifdef VFI_TEST_TAPS`) into a second library, libvfi_hip_test.so — the product objects + these — which
 * tests/conftest.py and tools/ select with cfi_amd._lib.use_test_build().  libvfi_hip.so exports exactly include/vfi_hip.h
 * (tests/test_capi_symbols.py::test_product_library_has_no_test_taps): nothing loaded beside a node can flip a model's kernel choice.
 * (Per-kernel tracing, vfi_trace_*, the clock probe and vfi_rife_work stay in vfi_hip.h: bench.py's roofline figures are part of
 * the deliverable, and none of them changes a result.)
 */
#ifndef VFI_HIP_TEST_H
#define VFI_HIP_TEST_H

#include "vfi_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Same contract as vfi_conv3x3, straightforward one-thread-per-output FMA kernel (cross-check of the MFMA path). */
int vfi_conv3x3_naive(const float* in_dev, const float* weight_host, const float* bias_host,
                      const float* beta_host, float* out_dev, int N, int H, int W, int Cin,
                      int Cout, int stride, int act, float slope, void* stream);


/* Algorithm choice of the 3x3 stride-1 layers (layer objects and the RIFE network): 0 automatic (Winograd where the launch has
 * enough work items), 1 direct implicit-GEMM kernel only, 2 Winograd wherever the shape allows; mode < 0 queries.  Returns the
 * mode in force.  Lets a test run BOTH forms of one layer / network on one input. */
int vfi_test_conv_algo(int mode);

/* The Winograd F(2x2,3x3) weight pack of a 3x3 layer (csrc/conv_wino.hip: U = G g G^T, layout
 * [Cout_p/32][Cin_p/8][j][xi/4][half][co%32][xi%4], physical input channel = c8*8 + half*4 + j), written to a HOST buffer: lets the
 * CPU suite check the pack and emulate the kernel's addressing (tests/test_wino_emulation.py).  Returns floats written or < 0. */
int64_t vfi_test_pack_wino3x3(const float* weight_host, int Cout, int Cin, const int* chan_map, int Cin_p, float* out_host, int64_t cap);
/* ConvTranspose2d(Cin, LO, 4, 2, 1) rewritten as ONE 3x3 stride-1 pad-1 convolution with 4 * LO output channels (channel g * LO + co =
 * output parity g = 2 py + px of channel co; csrc/conv_wino.hip: pack_deconv_as_conv3x3 — how the RIFE lastconv runs on the Winograd
 * kernel): w3_host OIHW [4 * LO][Cin][3][3], b3_host [4 * LO].  Host only.  Returns floats written to w3_host or < 0. */
int64_t vfi_test_pack_deconv3x3(const float* weight_host, const float* bias_host, int Cin, int LO, float* w3_host, float* b3_host, int64_t cap);

/* A/B options of tools/ and tests/: each selects between two CORRECT forms of a kernel or launch (csrc/vfi_common.h, enum Option):
 *   stage_quad (bit mask, default 14), fuse_encode (1), fuse0a (1), m2n2_px (-1), grouped_variant (-1), splitk (1), splat_atomic (0; 1 = LDS-atomic tile kernel, 3 = staged list gather with compaction for C == 4),
 *   splat_spill_cap (-1), wino_xcd (1), deconv_wino (1), encode_batched (1), wino_quant (1), m2m_fused (1: M2M render as one kernel), m2m_side (0; 1: M2M prepare forks its image-pyramid convolutions onto a side stream — measured neutral / slower under pair lanes), film_side (1: FILM forward on two streams — image 1's feature extraction and the backward flow pyramid beside image 0's / the forward one), xcd_bands (0), wino_probe (0; 1..4 = the cycle-ledger forms of the hot
 *   Winograd instantiation: same results, s_memtime stamps summed per wave of workgroup 0).
 * The product library reads NO experiment switch from the environment and does not contain this call: the defaults are all it can run.
 * Returns 0, or -2 for an unknown name. */
int vfi_test_set_option(const char* name, int64_t value);
/* The stamp sums of the last launch made under wino_probe != 0 on the current device: [4 waves][8] uint32 = q0, q1, q2, q3 (sums mod
 * 2^32 of the stamps the probe form takes, csrc/conv_wino.hip), stamps per sum, s_memtime at the wave's end, chunks, probe id.
 * Synchronises the device.  tools/wino_ledger.py turns them into the cycle ledger of docs/design/winograd.md. */
int vfi_test_wino_probe_read(uint32_t* out32);
/* Force direct-conv tile variants by trace name: "conv0a_b3=42,resconv_c128=36"; NULL / "" clears (tools/variant_sweep.sh). */
int vfi_test_variant_override(const char* spec);

/* Debug taps for parity tests: copy internal tensors of the LAST interpolate call to host.
 * what: 0 = flow after stage `stage` [B,Hp,Wp,4];  1 = stage input X of `stage`, planar4 [B,Cx/4,Hs,Ws,4];
 *       2 = frame slot pack, planar4 [2,Hp,Wp,4] = (rgb0 | encode features) (stage = slot).
 * Returns number of floats written or <0. */
int64_t vfi_rife_debug_read(vfi_rife_t* net, int what, int stage, float* host_buf, int64_t cap);
int vfi_rife_debug_keep(vfi_rife_t* net, int on);


/* The call order of FILM's greedy bisection for `inter_frames` new frames (film/__init__.py:17-40) as the C side computes it:
 * (left, right, new) grid positions, 3 ints per call; returns the number of calls or < 0.  Host only. */
int vfi_test_film_schedule(int inter_frames, int* triples, int cap);
/* ... and the float32 grid it is computed on: torch.linspace(0, 1, n) restated (n floats to `out`). */
int vfi_test_linspace01(int n, float* out);

/* FILM: the synthesised flow pyramid of the last vfi_film_forward — direction d (0 forward, 1 backward), pyramid level l,
 * [h_l, w_l, 2] floats to host.  Returns the number of floats or < 0. */
int64_t vfi_film_debug_read_flow(vfi_film_t* net, int d, int level, float* host_buf, int64_t cap);

/* M2M: internal tensors of the last vfi_m2m_prepare to host.  what 0: the PWC flows [2,Hp/4,Wp/4,2] (image 0 forward,
 * 1 backward); 1: d0 [2,Hp,Wp,8] = (refined-flow base 2 | normalised image 3 | warped partner 3); 2: r [2,Hp,Wp,12] = (8 flow
 * residuals | mask logit | pad).  Returns the number of floats or < 0. */
int64_t vfi_m2m_debug_read(vfi_m2m_t* net, int what, float* host_buf, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* VFI_HIP_TEST_H */
