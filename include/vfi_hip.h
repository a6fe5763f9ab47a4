/*
 * vfi_hip.h — C ABI of libvfi_hip.so, the MI355X (gfx950) frame-interpolation hot path.
 *
 * Drop-in boundary: this is what the reference's Python node layer would bind (ctypes) in
 * place of its torch.nn / cupy / taichi calls.  Plain pointers and sizes only — no torch
 * types.  All tensors are fp32.  "dev" pointers are device (HBM) addresses, e.g.
 * torch.Tensor.data_ptr() of a CUDA/HIP tensor; `stream` is a hipStream_t passed as void*
 * (NULL = the default stream).  Every function returns 0 on success, non-zero on error with
 * a message available from vfi_last_error().  Launches are asynchronous on `stream`.
 *
 * Layout convention: images and activations are NHWC ("channels last"), which is also the
 * layout of ComfyUI's IMAGE type ([N,H,W,C], reference vfi_utils.py:139-143) — the
 * reference's NHWC->NCHW einops views disappear on this path.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference
 * checkout, /root/reference in the build container).
 */
#ifndef VFI_HIP_H
#define VFI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------- */

/* Select the HIP device for the calling thread/process (one process per GPU).
 * Replaces comfy.model_management.get_torch_device() use at vfi_models/rife/__init__.py:121. */
int vfi_init(int device);
const char* vfi_last_error(void);
/* "gfx950" etc. of the active device, plus CU count. */
int vfi_device_info(char* arch_buf, int arch_buf_len, int* n_cus);

/* Per-kernel event tracing (bench.py roofline leg).  When enabled, every kernel launch made
 * by the library is bracketed by hipEventRecord on its stream.  vfi_trace_report() waits for
 * the recorded events and writes one line per kernel name: "<name> <calls> <total_ms>\n". */
int vfi_trace_enable(int on);
int vfi_trace_reset(void);
int vfi_trace_report(char* buf, int buf_len);

/* Shader-clock probe (bench.py `clock`, `roofline.frac_at_clock`).  The chip clocks to its power budget, so a launch duration alone
 * cannot tell a slower kernel from a slower box.  While a buffer is installed, every launch of the dominant (Winograd 3x3) kernel is
 * given one record of 8 x uint64 in it — { t0 = s_memtime at start, r0 = s_memrealtime at start, t1, r1 at end, tag, 0, 0, 0 } of
 * workgroup 0 (persistent: it lives as long as the launch) — until `capacity` records are used.  s_memtime ticks at the shader clock,
 * s_memrealtime at a constant rate (100 MHz; bench.py checks it against the HIP-event duration of the same launches):
 *   sustained MHz of launch i = (t1 - t0) / (r1 - r0) x 100.
 * dev_records: buffer of capacity * 8 uint64 on the CURRENT device, zeroed by the caller; NULL / 0 uninstalls (the names stay
 * readable).  Installing / uninstalling SYNCHRONISES the device first (a probed launch in flight re-reads the installed buffer for its
 * closing stamp), so the buffer may be freed as soon as the uninstalling call has returned.  At most `capacity` launches take a record
 * and a name.  vfi_clock_probe_names writes the trace names of the probed launches, one per line; line `tag` names record `tag`;
 * returns their count (< 0 on error). */
int vfi_clock_probe(void* dev_records, int capacity);
int vfi_clock_probe_names(char* buf, int buf_len);

/* ---- single-op entry points (parity tests, reuse by other nodes) ----------------------- */

/* RIFE backward warp: bilinear, padding_mode="border", align_corners=True, in the reference's
 * fp32 expression order (normalise -> grid_sample un-normalise round trip).
 * Replaces warp() at vfi_models/rife/rife_arch.py:31-70.
 *   in   [N,H,W,C]   flow [N,H,W,2] (x,y displacement in pixels)   out [N,H,W,C] */
int vfi_warp_border(const float* in_dev, const float* flow_dev, float* out_dev,
                    int N, int H, int W, int C, void* stream);

/* Generic NHWC convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 * Replaces torch.nn.Conv2d(k=3,pad=1,stride=1|2) (+ LeakyReLU / ResConv epilogue):
 *   vfi_models/rife/rife_arch.py:73-107 (conv), :20-28 (ResConv).
 *   weight: reference layout [Cout,Cin,3,3] (host pointer, repacked internally per call —
 *           test/utility entry; the model handles pre-pack at vfi_rife_create).
 *   in  [N,H,W,Cin]  out [N,Ho,Wo,Cout]   Ho = (H+2-3)/stride+1
 *   beta (host, [Cout]) may be NULL; if given: y = lrelu(conv*beta + in) (requires Cin==Cout, stride 1)
 *   act: 0 none, 1 LeakyReLU(slope)
 *   variant: -1 = heuristic, >=0 selects a tile configuration (see csrc/conv_mfma.hip); 100 / 101 = the Winograd F(2x2,3x3)
 *            form (stride 1; csrc/conv_wino.hip) with 16x8 / 32x4 output pixels per wave. */
int vfi_conv3x3(const float* in_dev, const float* weight_host, const float* bias_host,
                const float* beta_host, float* out_dev, int N, int H, int W, int Cin, int Cout,
                int stride, int act, float slope, int variant, void* stream);

/* ConvTranspose2d(Cin, Cout, 4, stride 2, pad 1) followed by PixelShuffle(2), output NHWC
 * [N,4H,4W,Cout/4].  Replaces IFBlock.lastconv, vfi_models/rife/rife_arch.py:215-218.
 *   weight: reference layout [Cin,Cout,4,4] (host). */
int vfi_deconv4x4_ps2(const float* in_dev, const float* weight_host, const float* bias_host,
                      float* out_dev, int N, int H, int W, int Cin, int Cout, void* stream);

/* ---- generic NHWC building blocks (FILM path; vfi_models/film/film_arch.py) -------------------------
 * Every tensor pointer addresses the first channel of a channel WINDOW inside a (possibly wider) NHWC tensor
 * whose pixel stride is `*_cs` floats, so the reference's torch.cat along channels is a matter of offsets. */

typedef struct vfi_conv vfi_conv_t;

/* nn.Conv2d(Cin, Cout, k, padding='same') with k in {1,2,3} (k=2 pads bottom/right like torch), weights
 * [Cout,Cin,k,k] in the reference layout (host).  `chan_map[ci]` (nullable) = position of reference input
 * channel ci inside the physical input window of `Cin_phys` channels (multiple of 8; unmapped positions get
 * zero weights — they must hold finite values).  Replaces film_arch.conv(), film_arch.py:784-798. */
vfi_conv_t* vfi_conv_create(const float* w_oihw_host, const float* bias_host, int Cout, int Cin, int kh, int kw,
                            const int* chan_map, int Cin_phys);
void vfi_conv_destroy(vfi_conv_t* conv);
/* out[..., :Cout] = act(conv(in[..., :Cin_phys]) + bias);  act: 0 none, 1 LeakyReLU(slope), 2 clamp to [0,1]
 * (the FILM node's prediction.clamp(0, 1), vfi_models/film/__init__.py:39). */
int vfi_conv_forward(const vfi_conv_t* conv, const float* in_dev, int in_cs, float* out_dev, int out_cs,
                     int N, int H, int W, int act, float slope, void* stream);
/* F.interpolate(x, scale_factor 2, 'nearest') + Conv2d(Cin, Cout, 2, padding 'same') — the first convolution of every FILM Fusion level
 * (film_arch.py:282-292) — as ONE layer that reads the LOW-resolution tensor (round 6): vfi_conv_forward(c, in [N,H,W,*], out [N,2H,2W,out_cs],
 * N, H, W, act 0 | 1, slope).  The four output parities are 4 * Cout channels of a 2x2 convolution whose weights are the original taps summed
 * per input pixel; parity (0,0) keeps 1 tap, (0,1) / (1,0) two, (1,1) four: 9 tap blocks instead of 16 and no up-sampled tensor.  Cout % 64 == 0.
 * Only for an exact x2 (the caller falls back to vfi_upsample_nearest + vfi_conv_forward otherwise). */
vfi_conv_t* vfi_conv_create_up2x2(const float* weight_oihw_host, const float* bias_host, int Cout, int Cin, const int* chan_map, int Cin_phys);

/* F.avg_pool2d(x, 2, 2) (odd sizes floor), film_arch.py:655-674, :112-113.  C % 4 == 0. */
int vfi_avgpool2(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int H, int W, int C, void* stream);
/* F.interpolate(x, size=(Hout,Wout), mode='nearest'), film_arch.py:286.  C % 4 == 0. */
int vfi_upsample_nearest(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int Hin, int Win,
                         int Hout, int Wout, int C, void* stream);
/* F.interpolate(mul * x, size=(Hout,Wout), mode='bilinear') (align_corners=False), film_arch.py:597,610,752. */
int vfi_resize_bilinear(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int Hin, int Win,
                        int Hout, int Wout, int C, float mul, void* stream);
/* film_arch.warp(image, flow_mul * flow): backward bilinear warp sampling at (x + fx, y + fy), border padding,
 * align_corners=False, reference fp32 expression order; flow = [dx, dy] per pixel.  film_arch.py:677-724. */
int vfi_warp_film(const float* in_dev, int in_cs, const float* flow_dev, int flow_cs, float flow_mul,
                  float* out_dev, int out_cs, int N, int H, int W, int C, void* stream);
/* out = alpha * a + beta * b over a C-channel window (b may be NULL): flow accumulation v = v_res + v,
 * multiply_pyramid, and channel-window copies.  film_arch.py:600-602,727-742. */
int vfi_axpby(const float* a_dev, int a_cs, const float* b_dev, int b_cs, float* out_dev, int out_cs,
              int64_t pixels, int C, float alpha, float beta, void* stream);

/* ---- FILM as one object ("same triplet for FILM") ------------------------------------------
 * The interpolator of the FILM node, weights resident, workspace owned, the launch sequence of one frame pair issued by
 * one call.  `tensors[i]` = host pointers to the 82 state_dict tensors in film_spec.film_shapes() order (== the state_dict
 * order of film_arch.Interpolator / of the TorchScript artifact film_net_fp32.pt), reference layouts; numels checked.
 * Replaces torch.jit.load(...) + model(frame0, frame1, dt) of vfi_models/film/__init__.py:73-76,30-39 and
 * Interpolator.debug_forward, film_arch.py:401-455 (the model ignores dt: always the midpoint, :427-429). */
typedef struct vfi_film vfi_film_t;
vfi_film_t* vfi_film_create(const float* const* tensors, const int64_t* numels, int n_tensors);
void vfi_film_destroy(vfi_film_t* net);
/* out_dev [H,W,3] = Interpolator(x0, x1) for x0_dev, x1_dev [H,W,C] fp32 (C >= 3, alpha ignored); clamp != 0 applies the
 * node's prediction.clamp(0, 1) (film/__init__.py:39).  H, W >= 64.  The workspace (15 GB at 1080p) is sized on first use.
 * All work is ordered on `stream` as seen by the caller; inside, half of the network up to the fusion runs on a side stream of the
 * object's own (forked from and joined to `stream` by events within the call: round 6, docs/design/film.md). */
int vfi_film_forward(vfi_film_t* net, const float* x0_dev, const float* x1_dev, int C, int H, int W, float* out_dev, int clamp,
                     void* stream);
int vfi_film_release_workspace(vfi_film_t* net);
/* on != 0 (the default): vfi_film_forward forks half of the network onto the object's side stream; 0: everything on the caller's stream —
 * what a host that already keeps several pairs in flight on streams of its own (the nodes' pair lanes) wants: four busy streams on the
 * runtime's four hardware queues left the lanes 2 % slower than two.  Frames are bit-identical either way.  Returns the previous setting. */
int vfi_film_two_streams(vfi_film_t* net, int on);

/* The whole FILM node call for a HOST clip, replacing FILM_VFI.vfi's body, vfi_models/film/__init__.py:63-113:
 * frames_host [N,H,W,C] fp32 (C >= 3; alpha dropped) -> out_host [*n_out,H,W,3].  Per kept pair: frame_i, then the m-1 new frames
 * of the greedy bisection (:12-42; every model call returns the midpoint of two known frames, clamped to [0,1], and later calls
 * consume earlier outputs); a skipped pair is DROPPED, frame included (:89-90); the clip's last frame is appended.
 * multipliers == NULL: `multiplier` for every pair; else the list (n_multipliers entries) padded with 2 (:84-87).  skip: [N-1]
 * flags or NULL.  out_host == NULL only computes *n_out.  Synchronous, own stream. */
int vfi_film_run(vfi_film_t* net, const float* frames_host, int N, int H, int W, int C, int multiplier, const int* multipliers,
                 int n_multipliers, const uint8_t* skip, float* out_host, int64_t* n_out);

/* ---- M2M custom ops ---------------------------------------------------------------------- */

/* Summation splat (forward warp): out[n, y', x', c] += in[n,y,x,c] * bilinear weight at the 4 integer
 * neighbours of (x + flow_x, y + flow_y); out-of-image targets dropped, non-finite targets skipped.
 * `out` is zeroed by the call.  Replaces softsplat_func.forward / the softsplat_out CUDA kernel,
 * vfi_models/ops/cupy_ops/softsplat.py:140-233.   in/out [N,H,W,C], flow [N,H,W,2] (NHWC). */
int vfi_softsplat_sum(const float* in_dev, const float* flow_dev, float* out_dev, int N, int H, int W, int C,
                      void* stream);

/* 9x9 mean-L1 cost volume: out[n,y,x, out_coff + 9*(dy+4)+(dx+4)] = mean_c |one[n,y,x,c] - two[n,y+dy,x+dx,c]|,
 * out-of-image (y+dy,x+dx) -> mean_c |one|.  Replaces costvol_func.forward / costvol_out,
 * vfi_models/ops/cupy_ops/costvol.py:4-43,135-183.   one/two [N,H,W,C] (C % 4 == 0); out [N,H,W,out_cs]:
 * the 81 channels are written at channel offset out_coff of a wider NHWC tensor (the reference's
 * torch.cat of the volume with features, vfi_models/m2m/M2M_arch.py:484-490, then costs nothing). */
int vfi_costvol9x9(const float* one_dev, int one_cs, const float* two_dev, int two_cs, int two_swap, float* out_dev,
                   int N, int H, int W, int C, int out_cs, int out_coff, void* stream);
/* one/two are channel windows (pixel strides one_cs/two_cs); two_swap != 0 reads `two` from the partner image
 * n^1 of a batch of direction pairs (the reference runs the decoder on (a,b) and on (b,a), M2M_arch.py:521-546). */

/* ---- M2M network building blocks (vfi_models/m2m/M2M_arch.py) -------------------------------- */

/* Generalised layer object: kind 0 = nn.Conv2d(Cin, Cout, k, stride, padding) with (k,stride) in
 * {(3,1),(3,2),(2,2),(1,1)}: k=3 pads 1 — pad_mode 0 zeros (M2M_arch.py:589-602), 1 replicate (the "conv(3,replpad)"
 * of Basic, :228-260); k=2/s=2 pads 0 ("sconv(2)" after evenize, :205-226 — sizes must be even).
 * kind 1 = nn.ConvTranspose2d(Cin, Cout, 4, 2, 1) (deconv(), :605-618), weights [Cin,Cout,4,4].
 * prelu_host (nullable): Cout per-channel PReLU slopes, applied by act 3. */
vfi_conv_t* vfi_conv_create_ex(int kind, const float* w_host, const float* bias_host, int Cout, int Cin, int k, int stride,
                               int pad_mode, const int* chan_map, int Cin_phys, const float* prelu_host);
/* out = post_scale * act(layer(in) + bias + res) + post_shift.  act: 0 none, 1 LeakyReLU / single-parameter PReLU
 * (slope), 2 clamp01, 3 per-channel PReLU, 4 sigmoid, 5 GELU (erf form, nn.GELU()).  post_scale == 0 disables the affine.  res (nullable):
 * [N,Hout,Wout,res_cs] added before the activation (the decoder's `flow + netMain(...)`, :503).
 * Output is [N, Hin/stride, Win/stride, out_cs] (kind 0) or [N, 2*Hin, 2*Win, out_cs] (kind 1). */
int vfi_conv_forward_ex(const vfi_conv_t* conv, const float* in_dev, int in_cs, int Hin, int Win, float* out_dev, int out_cs,
                        int N, int act, float slope, float post_scale, float post_shift, const float* res_dev, int res_cs,
                        void* stream);

/* Replicate-pad both frames to Hp x Wp (M2M_arch.py:903-913), joint mean/std over the padded pair (:915-931),
 * and write (frame_n - mean) / (std + 1e-7) to out[n, y, x, out_coff + 0..2] (n = 0,1; out [2,Hp,Wp,out_cs]).
 * stats_dev[0] = mean, stats_dev[1] = std + 1e-7.  frames: [H,W,C] fp32, C >= 3.  workspace: >= 16 KiB device. */
int vfi_m2m_normalize(const float* frame0_dev, const float* frame1_dev, int C, int H, int W, int Hp, int Wp, float* out_dev,
                      int out_cs, int out_coff, float* stats_dev, void* workspace_dev, int64_t workspace_bytes, void* stream);
/* backwarp(): grid_sample(in, grid + flow*2/(size-1), bilinear, zeros, align_corners=True), M2M_arch.py:24-92.
 * in_swap != 0 samples the partner image n^1.  flow = [dx, dy] per pixel. */
int vfi_warp_m2m(const float* in_dev, int in_cs, int in_swap, const float* flow_dev, int flow_cs, float* out_dev, int out_cs,
                 int N, int H, int W, int C, void* stream);
/* adaptive_avg_pool2d to 1x1 (mode 0 -> out [N,1,C]), Hx1 (mode 1 -> [N,H,C]) or 1xW (mode 2 -> [N,W,C]), :689-703 */
int vfi_pool_mean(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int H, int W, int C, int mode, void* stream);
/* out = s3 * mean_k(cC[n,k*C+c] * cH[n,y,k] * cW[n,x,k]), k < 16 — the EncDec "cube" attention, :786-795 */
int vfi_m2m_cube_apply(const float* s3_dev, int s3_cs, const float* cC_dev, const float* cH_dev, int cH_cs, const float* cW_dev,
                       int cW_cs, float* out_dev, int out_cs, int N, int H, int W, int C, void* stream);
/* Timestep-independent splat preparation for the 8 splats s = 2*branch + direction of one pair (:945-1010, :559-561):
 * tf_dev [8,H,W,2] = flow + residual, e_dev [8,H,W] = exp(clip(alpha * photometric, -20, 20)).
 * d0 [2,H,W,d0_cs] = (flow 0..1 | normalised image 2..4 | ..), r [2,H,W,r_cs] = (8 residuals | mask logit). */
int vfi_m2m_photo(const float* d0_dev, int d0_cs, const float* r_dev, int r_cs, float alpha, float* tf_dev, float* e_dev, int H,
                  int W, void* stream);
/* img4_dev [2,H,W,4] = (d0's channels 2..4 = the normalised image, 1): the compact plane the kernels below read with one 16-byte load
 * per tap / source (round 6, csrc/m2m_render.hip). */
int vfi_m2m_image4(const float* d0_dev, int d0_cs, float* img4_dev, int H, int W, void* stream);
/* = vfi_warp_m2m(image, in_swap = 1, C = 3) with the partner image taken from img4 (bit-identical): out[n,y,x,0..2] =
 * backwarp(img4[n ^ 1], flow[n]) (M2M_arch.py:24-92, :866-890). */
int vfi_m2m_warp_image4(const float* img4_dev, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int H, int W, void* stream);
/* vfi_m2m_photo's results computed per 32x32 tile, plus what the one-kernel render below needs: img4_dev from vfi_m2m_image4
 * (input); tile_ranges_dev [8][ceil(H/32)*ceil(W/32)][4] =
 * (fx_min, fx_max, fy_min, fy_max) of every refined flow field tf_s over the tile's finite values (min > max: none);
 * smax_dev [8] = max |tf_s|.  tf_dev / e_dev as vfi_m2m_photo (bit-identical). */
int vfi_m2m_photo_tiles(const float* d0_dev, int d0_cs, const float* r_dev, int r_cs, float alpha, const float* img4_dev, float* tf_dev,
                        float* e_dev, float* tile_ranges_dev, float* smax_dev, int H, int W, void* stream);
/* forwarp_mframe_mask for one timestep 0 <= t <= 1 as ONE kernel (M2M_arch.py:551-581, :1012-1037 with softsplat_out of
 * cupy_ops/softsplat.py:140-192 inside): = vfi_m2m_splat_inputs + vfi_softsplat_sum (8 splats) + vfi_m2m_combine, bit-identical to
 * that sequence wherever a tile's source window fits one LDS stage (1824 sources), same sums in another strip order beyond; no limit
 * on the displacement, no fallback launches.  Inputs from vfi_m2m_photo_tiles (Hp x Wp = the padded size), stats from
 * vfi_m2m_normalize; out_dev [H,W,3]. */
int vfi_m2m_render_fused(const float* img4_dev, const float* tf_dev, const float* e_dev, const float* tile_ranges_dev, const float* smax_dev,
                         const float* stats_dev, float t, float* out_dev, int Hp, int Wp, int H, int W, void* stream);
/* Per timestep t: in_dev [8,H,W,4] = (image*td*e, td*e), flow_dev [8,H,W,2] = tf * tm (:1012-1024, :563-567) */
int vfi_m2m_splat_inputs(const float* d0_dev, int d0_cs, const float* tf_dev, const float* e_dev, float t, float* in_dev,
                         float* flow_dev, int H, int W, void* stream);
/* splat_dev [8,Hp,Wp,4] (vfi_softsplat_sum of the above) -> out [H,W,3]: normalise by the splatted weights, fill holes
 * with the t-blend, undo the input normalisation, crop (:569-581, :1026-1037) */
int vfi_m2m_combine(const float* splat_dev, const float* d0_dev, int d0_cs, const float* stats_dev, float t, float* out_dev,
                    int Hp, int Wp, int H, int W, void* stream);

/* ---- M2M as one object ("same triplet for M2M") ---------------------------------------------
 * `tensors[i]` = host pointers to the 188 state_dict tensors of M2M_PWC in m2m_spec.m2m_shapes() order (== torch state_dict
 * order), reference layouts; numels checked.  Replaces M2M_PWC() + load_state_dict (vfi_models/m2m/__init__.py:43-46) and
 * M2M_PWC.forward (M2M_arch.py:894-1037), split where the reference's timestep loop starts (:948):
 *   vfi_m2m_prepare — padding, normalisation, flow network, motion refinement, photometric metric: once per frame pair;
 *   vfi_m2m_render  — the 8 splats + blend of ONE timestep t -> out_dev [H,W,3] (not clamped, like the reference). */
typedef struct vfi_m2m vfi_m2m_t;
vfi_m2m_t* vfi_m2m_create(const float* const* tensors, const int64_t* numels, int n_tensors);
void vfi_m2m_destroy(vfi_m2m_t* net);
int vfi_m2m_prepare(vfi_m2m_t* net, const float* frame0_dev, const float* frame1_dev, int C, int H, int W, void* stream);
int vfi_m2m_render(vfi_m2m_t* net, float t, float* out_dev, void* stream);
int vfi_m2m_release_workspace(vfi_m2m_t* net);
/* device bytes of the object's activation tensors at the moment (0 after vfi_m2m_release_workspace): what a host that keeps the
 * object between clips weighs against re-allocating per clip (the nodes: ckpt.end_call) */
int64_t vfi_m2m_workspace_bytes(vfi_m2m_t* net);

/* The whole M2M node call for a HOST clip, replacing M2M_VFI.vfi + generic_frame_loop in timestep mode
 * (vfi_models/m2m/__init__.py:33-60, vfi_utils.py:149-389): frames_host [N,H,W,C] fp32, N >= 2 -> out_host [*n_out,H,W,3].
 * multipliers == NULL ("int multiplier"): frame_i, its multiplier-1 new frames at t = k/m (none when skip[i]), ..., last frame.
 * multipliers != NULL ("list multiplier", n_multipliers entries padded with 2): every pair runs as its own 2-frame loop — m == 0
 * drops the pair INCLUDING its first frame (and the clip's last frame when it is the last pair), m == 1 keeps the frame, and the
 * skip list is consulted with the pair's LOCAL index, i.e. skip[0], for every pair (vfi_utils.py:364-386): reproduced as is.
 * New frames are not clamped (like the reference).  out_host == NULL only computes *n_out.  Synchronous, own stream. */
int vfi_m2m_run(vfi_m2m_t* net, const float* frames_host, int N, int H, int W, int C, int multiplier, const int* multipliers,
                int n_multipliers, const uint8_t* skip, float* out_host, int64_t* n_out);

/* ---- RIFE arch 4.0 building blocks (sudo_rife4 checkpoint; rife40.py drives them with the layer objects above) -- */

/* Block-0 input of one task: out [Hp,Wp,8] = (clamp(frame0.rgb), clamp(frame1.rgb), timestep, 0), images zero in the
 * padding, timestep everywhere (torch.clamp + F.pad + timestep.repeat, rife_arch.py:476-499).  frames [H,W,C>=3]. */
int vfi_rife40_prep(const float* frame0_dev, const float* frame1_dev, int C, int H, int W, float timestep, float* out_dev,
                    int Hp, int Wp, void* stream);
/* warp() (rife_arch.py:31-70: bilinear, border, align_corners=True, reference expression order) over NHWC channel
 * windows; flow = [dx, dy] at flow_dev[pixel * flow_cs]. */
int vfi_warp_rife(const float* in_dev, int in_cs, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N,
                  int H, int W, int C, void* stream);
/* *out_dev = max |x| over a C-channel window of `pixels` pixels — the `f0[:, :2].abs().max() > 32` test that doubles
 * the block scales (rife_arch.py:598-607). */
int vfi_absmax(const float* x_dev, int cs, int C, int64_t pixels, float* out_dev, void* stream);
/* merged = w0 * sigmoid(mask) + w1 * (1 - sigmoid(mask)); with res (the Unet output): clamp(merged + (res*2 - 1), 0, 1);
 * crop to H x W and the node's clamp(0,1) (rife_arch.py:712-732, rife/__init__.py:207).  w01 [B,Hp,Wp,>=6] = (w0 | w1). */
int vfi_rife40_output(const float* w01_dev, int w_cs, const float* mask_dev, int m_cs, const float* res_dev, int r_cs,
                      float* out_dev, int B, int Hp, int Wp, int H, int W, void* stream);

/* ---- IFRNet building blocks (IFRNet_L / IFRNet_S checkpoints; ifrnet.py drives them with the layer objects above) -- */

/* img{0,1} [Hp,Wp,4] = (frame{0,1}.rgb, 0), zero in the padding: F.pad of both frames, no clamp
 * (vfi_models/ifrnet/IFRNet_L_arch.py:230-235).  frames [H,W,C>=3]. */
int vfi_ifrnet_prep(const float* frame0_dev, const float* frame1_dev, int C, int H, int W, float* img0_dev, float* img1_dev,
                    int Hp, int Wp, void* stream);
/* mean_[n] = mean over both padded images of pair n (IFRNet_L_arch.py:242-247), from the per-image channel means
 * chan_means [2N,4] (vfi_pool_mean mode 0 over img [2N,Hp,Wp,4]; images 0..N-1 = img0, N..2N-1 = img1);
 * writes mean_out[n] and subtracts it from the colour channels of both images in place (:248-249). */
int vfi_ifrnet_center(float* img_dev, const float* chan_means_dev, float* mean_out_dev, int N, int64_t pixels, void* stream);
/* convrelu(3, 64, 7, 2, 3): Conv2d 7x7 stride 2 pad 3 + PReLU(64), the head of IFRNet_L's encoder (:128-130).
 * w_dev [7][7][3][Cout] (device), out [N,(Hin-1)/2+1,(Win-1)/2+1,out_cs]. */
int vfi_conv7x7s2_prelu(const float* in_dev, int in_cs, const float* w_dev, const float* bias_dev, const float* slope_dev, int Cout,
                        float* out_dev, int out_cs, int N, int Hin, int Win, void* stream);
/* torch.sigmoid in place over a C-channel window (:278) */
int vfi_sigmoid(float* x_dev, int cs, int C, int64_t pixels, void* stream);
/* out[n, p, 0..C) = values_host[n] for N <= 64 items of pixels_per_item pixels: embt.repeat(1, 1, h, w) (:160-163) */
int vfi_fill_items(float* out_dev, int cs, int C, int N, int64_t pixels_per_item, const float* values_host, void* stream);
/* F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False) * post_mul with the source step the CALLER
 * derived from s (torch uses (float)(1/s), not Hin/Hout, when a scale_factor is given) and the output size floor(in*s)
 * (:38-41,250-251,281-288). */
int vfi_resize_bilinear_ratio(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int Hin, int Win, int Hout, int Wout,
                              int C, float ratio_h, float ratio_w, float post_mul, void* stream);
/* imgt = clamp(mask * warp(img0, flow0) + (1 - mask) * warp(img1, flow1) + mean_ + res, 0, 1)[:H, :W] (:290-294).
 * fin [N,Hf,Wf,8] = (flow0 xy, flow1 xy, mask, res rgb) at the size of the reference's last resize (the warp grid has
 * that size, the images Hp x Wp); img{0,1} [N,Hp,Wp,4] mean-removed; out [N,H,W,3]. */
int vfi_ifrnet_output(const float* img0_dev, const float* img1_dev, const float* fin_dev, const float* mean_dev, float* out_dev, int N,
                      int Hp, int Wp, int Hf, int Wf, int H, int W, void* stream);

/* ---- GMFSS Fortuna (union) building blocks (vfi_models/gmfss_fortuna/GMFSS_Fortuna_union_arch.py; gmfss.py drives them
 * together with the layer objects above).  First-correct versions: one thread per output element. ------------------------ */

/* F.pad of an RGB frame into channels 0..2 of an NHWC tensor, zeros in the padding (gmfss_fortuna/__init__.py:43-48) */
int vfi_pad_rgb(const float* frame_dev, int C, int H, int W, float* out_dev, int out_cs, int Hp, int Wp, void* stream);
/* (x - mean[c]) / std[c], C <= 8: normalize_img (:1123-1131) */
int vfi_normalize_channels(const float* in_dev, int in_cs, float* out_dev, int out_cs, int C, int64_t pixels, const float* mean_host,
                           const float* std_host, void* stream);
/* nn.PReLU() (one shared slope) over a channel window: the pre-activations of MetricNet / FeatureNet / GridNet (:1420-1561) */
int vfi_prelu_scalar(const float* in_dev, int in_cs, float* out_dev, int out_cs, int C, int64_t pixels, float slope, void* stream);
/* nn.InstanceNorm2d (no affine, eps 1e-5): stats [N][C][2] = (mean, 1/sqrt(var+eps)); workspace >= N*64*C*2 doubles
 * (:165-215); a larger one is used for more, shorter strips of the first pass (up to 1024: more workgroups in flight) */
int vfi_instnorm_stats(const float* x_dev, int cs, int C, int N, int64_t HW, float* stats_dev, double* workspace_dev,
                       int64_t workspace_bytes, void* stream);
/* out = act2(act1((x - mean) * rstd) + add): norm(+relu) and the residual sum + relu of ResidualBlock_class (:207-215) */
int vfi_instnorm_apply(const float* x_dev, int cs, const float* stats_dev, int C, int N, int64_t HW, int relu1, const float* add_dev,
                       int add_cs, int relu2, float* out_dev, int out_cs, void* stream);
/* nn.LayerNorm(C) over the channel axis (eps 1e-5), TransformerLayer.norm1 / norm2 (:479-523) */
int vfi_layernorm(const float* x_dev, int cs, int C, int64_t tokens, const float* gamma_dev, const float* beta_dev, float* out_dev,
                  int out_cs, void* stream);
/* out = add + LayerNorm(x) (`source + message`, TransformerLayer.forward :517-523), out2 (nullable) = a second copy of the result in
 * another channel window (the FFN's concat slot).  out may be add itself (in place); neither output may alias x. */
int vfi_layernorm_add(const float* x_dev, int cs, int C, int64_t tokens, const float* gamma_dev, const float* beta_dev, const float* add_dev,
                      int add_cs, float* out_dev, int out_cs, float* out2_dev, int out2_cs, void* stream);
/* nn.GELU() (erf form) in place over a channel window (:467-471) */
int vfi_gelu(float* x_dev, int cs, int C, int64_t pixels, void* stream);
/* torch.roll by (-shift_h, -shift_w) + split_feature into splits x splits windows: [B,h,w,C] -> [B*K*K, (h/K)*(w/K), C];
 * inverse != 0: merge_splits + roll back (:367-436,1059-1120) */
int vfi_window_partition(const float* in_dev, int in_cs, float* out_dev, int out_cs, int B, int h, int w, int C, int splits, int shift_h,
                         int shift_w, int inverse, void* stream);
/* out[b][m][n] = alpha * sum_k A[b][m][k] * B[b][n][k]: q k^T / sqrt(c) of the attention and matching steps (:319,417,810) */
int vfi_bmm_nt(const float* a_dev, int a_cs, const float* b_dev, int b_cs, float* out_dev, int nb, int M, int N, int K, float alpha,
               void* stream);
/* out[b][m][c] = sum_n P[b][m][n] * V[b][n][c]: attn v, prob grid, prob flow (:321,425,833,741) */
int vfi_bmm_nn(const float* p_dev, const float* v_dev, int v_cs, float* out_dev, int out_cs, int nb, int M, int N, int C, void* stream);
/* softmax over the rows of x [nb][rows][cols] in place; mask [period][rows][cols] (nullable) is added first, batch b taking
 * mask[b % period] (scores += attn_mask.repeat(b, 1, 1), :422-425) */
int vfi_softmax_rows(float* x_dev, int nb, int rows, int cols, const float* mask_dev, int mask_period, void* stream);
/* The three calls above as ONE flash-style kernel on the fp32 matrix cores, the score matrix never touching HBM:
 *   out[b][m][0:DV] = sum_n softmax_n(alpha * q[b][m] . k[b][n] + mask) * v[b][n][0:DV],   q, k: [nb][L][C = 128]
 * mask = -100 where labels[b % period][m] != labels[b % period][n] (labels nullable, int32 [period][Lk]: the shifted-window
 * attention mask of generate_shift_window_attn_mask, :326-364, as region labels).  DV = 128 (attention) or 1..32 (global
 * matching / flow propagation: v = pixel grid / flow).  Replaces single_head_split_window_attention (:367-436),
 * global_correlation_softmax (:806-843), FeatureFlowAttention.forward's global branch (:708-745). */
int vfi_attention(const float* q_dev, int q_cs, const float* k_dev, int k_cs, const float* v_dev, int v_cs, float* out_dev,
                  int out_cs, int nb, int Lq, int Lk, int C, int DV, float alpha, const int* labels_dev, int label_period,
                  void* stream);
/* single_head_split_window_attention (:367-436) on [B,h,w,.] token MAPS, with the roll + split into splits x splits windows
 * (split_feature / merge_splits, :1059-1120) and the roll back folded into the kernel's addressing — no partitioned copies
 * of q / k / v / out in HBM.  shift = (0,0) or (wh/2, ww/2) with labels_dev = int32 [splits^2][wh*ww] region labels (nullable).
 * out must not alias q / k / v. */
int vfi_window_attention(const float* q_dev, int q_cs, const float* k_dev, int k_cs, const float* v_dev, int v_cs, float* out_dev,
                         int out_cs, int B, int h, int w, int splits, int shift_h, int shift_w, int C, float alpha,
                         const int* labels_dev, void* stream);
/* flow_warp / bilinear_sample: grid_sample(zeros, align_corners=True) at pixel + flow (:955-991) */
int vfi_flow_sample(const float* in_dev, int in_cs, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N, int H, int W,
                    int C, void* stream);
/* F.interpolate(x, size=(Hout,Wout), mode="bilinear", align_corners=True) * post_mul: the x2 flow up-sampling between GMFlow's
 * scales (:1302-1307) */
int vfi_resize_bilinear_ac(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int Hin, int Win, int Hout, int Wout,
                           int C, float post_mul, void* stream);
/* flow[:, 0:2] += local_correlation_softmax(f0, f1, radius) (:846-913,1335) */
int vfi_local_match(const float* f0_dev, int f0_cs, const float* f1_dev, int f1_cs, float* flow_dev, int flow_cs, int N, int H, int W,
                    int C, int radius, void* stream);
/* FeatureFlowAttention.forward_local_window_attn: q / k already projected, (2r+1)^2 window, zero padding (:745-803) */
int vfi_local_propagate(const float* q_dev, int q_cs, const float* k_dev, int k_cs, const float* flow_dev, int flow_cs, float* out_dev,
                        int out_cs, int N, int H, int W, int C, int radius, void* stream);
/* convex up-sampling of the flow by `factor` from the up-sampler's 9*factor^2 mask logits (:1237-1258) */
int vfi_convex_upsample(const float* mask_dev, int mask_cs, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N, int H,
                        int W, int factor, void* stream);
/* MetricNet's 14 input channels: images, -l1 photometric errors after backwarp, normalised flows, fwd/bwd occlusion (:1375-1454) */
int vfi_gmfss_metric_inputs(const float* img0_dev, const float* img1_dev, int img_cs, const float* flow01_dev, const float* flow10_dev,
                            int flow_cs, float* out_dev, int out_cs, int H, int W, void* stream);
/* tanh(x) * scale in place over a channel window (:1465) */
int vfi_tanh_scale(float* x_dev, int cs, int C, int64_t pixels, float scale, void* stream);
/* softsplat(x, flow, metric, "soft"), the part before the summation splat: out [px, C+1] = (x * exp(zs*z), exp(zs*z)),
 * flow_out [px, 2] = fs * flow (vfi_models/ops/cupy_ops/softsplat.py:408-409) */
int vfi_splat_prep(const float* x_dev, int x_cs, const float* z_dev, int z_cs, const float* flow_dev, int flow_cs, float* out_dev,
                   float* flow_out_dev, int C, int64_t pixels, float z_scale, float flow_scale, void* stream);
/* ... and after it: out[p, 0:C] = splat[p, 0:C] / (splat[p, C] + 1e-7) (softsplat.py:415-432) */
int vfi_splat_normalize(const float* splat_dev, float* out_dev, int out_cs, int C, int64_t pixels, void* stream);
/* nn.PixelShuffle(2) on NHWC: in [N,H,W,4C] -> out [N,2H,2W,C] (IFBlock.lastconv rife_arch.py:215-218, GridNet tail :1564-1579) */
int vfi_pixel_shuffle2(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int H, int W, int C, void* stream);
/* torch.clamp(x, 0, 1)[:, :, :H, :W] into a dense [H,W,C] frame (:1857, gmfss_fortuna/__init__.py:78) */
int vfi_clamp_crop(const float* in_dev, int in_cs, int Hp, int Wp, float* out_dev, int H, int W, int C, void* stream);

/* ---- IFUNet building blocks (vfi_models/ifunet/IFUNet_arch.py; ifunet.py drives them with the layer objects above).
 * Bodies checked on the host (tests/hostcheck) and on the MI355X against the oracle (tests/test_gpu_ifunet.py). --------------- */

/* CBAM ChannelGate pooling (:411-436): stats [N][C][2] = (mean, max) over H*W; workspace >= N*64*C*12 bytes
 * (a larger one is used for more, shorter strips of the first pass, up to 128) */
int vfi_channel_pool(const float* x_dev, int cs, int C, int N, int64_t HW, float* stats_dev, void* workspace_dev, int64_t workspace_bytes,
                     void* stream);
/* scale [N][C] = sigmoid(mlp(mean) + mlp(max)), mlp = Linear(C,R) -> ReLU -> Linear(R,C); w1 [R][C], w2 [C][R] on the device (:447-452) */
int vfi_cbam_gate(const float* stats_dev, const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* b2_dev, int C, int R,
                  int N, float* scale_dev, void* stream);
/* xs = x * scale[n][c] and ChannelPool comp [N*HW][2] = (max_c xs, mean_c xs) (:451-466) */
int vfi_cbam_scale_compress(const float* x_dev, int cs, const float* scale_dev, int C, int N, int64_t HW, float* xs_dev, int xs_cs,
                            float* comp_dev, void* stream);
/* SpatialGate in place: xs *= sigmoid(bn_a * conv7x7(comp) + bn_b); w [7][7][2] on the device, BatchNorm folded (:469-482) */
int vfi_cbam_spatial(float* xs_dev, int cs, const float* comp_dev, const float* w_dev, float bn_a, float bn_b, int C, int N, int H, int W,
                     void* stream);
/* IFBlock.upsample_flow: convex up-sampling by `factor` of a flow with up to 8 channels (:627-638) */
int vfi_convex_upsample_c(const float* mask_dev, int mask_cs, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N, int H,
                          int W, int factor, int flow_channels, void* stream);
/* out = a * mask + b * (1 - mask), mask one channel (:764) */
int vfi_lerp_mask(const float* a_dev, int a_cs, const float* b_dev, int b_cs, const float* mask_dev, int mask_cs, float* out_dev,
                  int out_cs, int C, int64_t pixels, void* stream);
/* out = clamp(a + b, 0, 1) (:161) */
int vfi_add_clamp01(const float* a_dev, int a_cs, const float* b_dev, int b_cs, float* out_dev, int out_cs, int C, int64_t pixels,
                    void* stream);
/* ResynNet's blend, cropped to H x W: softmax over (clamp(m0,-4,4), clamp(m1,-4,4), 0) weights img0, img1, deg (:188-192,766) */
int vfi_ifunet_blend(const float* img0_dev, const float* img1_dev, const float* deg_dev, int img_cs, const float* mask0_dev,
                     const float* mask1_dev, int mask_cs, float* out_dev, int Hp, int Wp, int H, int W, void* stream);
/* out[p, 0:C] = value (timestep planes, :667-670) */
int vfi_fill_channels(float* out_dev, int cs, int C, int64_t pixels, float value, void* stream);

/* ---- RIFE 4.7 / 4.9 model --------------------------------------------------------------- */

typedef struct vfi_rife vfi_rife_t;

/* Build the device-resident network from a reference checkpoint.
 * arch_ver_x10: 47 = architecture "4.7" (rife47.pth / rife49.pth, 124 tensors), 417 = "4.17" (rife417.pth, 128
 * tensors: same IFBlocks on 8 feature channels per frame, encoder Head_417, rife_arch.py:355-375,417-433),
 * 426 = "4.26" (rife426.pth, 158 tensors: 5 IFBlocks with carried block features, encoder Head, :378-398,451-457).
 * `tensors[i]` are host pointers to the state_dict tensors in the key order of rife_spec.rife_shapes(arch)
 * (== torch state_dict order of IFNet(arch)), each in the reference's own layout; `numels[i]` their element
 * counts (checked).  Replaces IFNet(arch_ver).load_state_dict(torch.load(path)) + .to(device),
 * vfi_models/rife/__init__.py:129-135. */
vfi_rife_t* vfi_rife_create(int arch_ver_x10 /* 47 | 417 | 426 */, const float* const* tensors,
                            const int64_t* numels, int n_tensors);
void vfi_rife_destroy(vfi_rife_t* net);

/* (Re)size the workspace: frames of H x W (unpadded), up to `max_batch` tasks per forward and
 * `n_slots` cached frames.  scale_factor as in the node widget (scale_list = [8,4,2,1]/sf,
 * vfi_models/rife/__init__.py:157-160): 0.25, 0.5, 1, 2, 4 — block scales >= 1 must divide the padded size,
 * block scales 0.5 / 0.25 (sf 2 / 4) run the block at 2x / 4x the frame resolution (rife_arch.py:237-276). */
int vfi_rife_configure(vfi_rife_t* net, int H, int W, int max_batch, int n_slots, float scale_factor);

/* Upload-side half of IFNet.forward for ONE input frame: clamp to [0,1], zero-pad to x64
 * (rife_arch.py:476-484) and `encode` (rife_arch.py:414-416,501-503) into frame slot `slot`.
 * frame_dev: [H,W,C] fp32, C>=3 (alpha dropped, vfi_utils.py:139-140).
 * The result depends only on the frame, so it is computed once per frame and shared by both
 * adjacent pairs and all timesteps. */
int vfi_rife_load_frame(vfi_rife_t* net, int slot, const float* frame_dev, int C, void* stream);

/* 8-bit frames (the callers either side of the node — video load / save — deal in uint8 images, a
 * quarter of the PCIe bytes): the same as vfi_rife_load_frame on frame_dev[H,W,C] uint8 with x = u8 / 255 computed on the
 * device (== torch's ``frames.float() / 255`` bit for bit), and the way back: out = round(clamp(in, 0, 1) * 255), ties to even. */
int vfi_rife_load_frame_u8(vfi_rife_t* net, int slot, const uint8_t* frame_dev, int C, void* stream);
/* The same for n frames at once: frames_dev[i] ([H,W,C] fp32, or uint8 when is_u8) -> slot slots[i] (distinct).  ONE launch for the
 * whole list on arch 4.7 (what a batch of the node's loop, rife/__init__.py:195-207, needs resident: 2 frames per task); results are
 * bit-identical to n single calls. */
int vfi_rife_load_frames(vfi_rife_t* net, int n, const int* slots, const void* const* frames_dev, int C, int is_u8, void* stream);
int vfi_f32_to_u8(const float* in_dev, uint8_t* out_dev, int64_t n, void* stream);

/* The per-task hot loop: out[b] = clamp(IFNet(frame[slot0[b]], frame[slot1[b]], t[b]), 0, 1)
 * for b < B.  Replaces the model call + clamp at vfi_models/rife/__init__.py:200-207 and
 * IFNet.forward (rife_arch.py:465-732, arch "4.7").   out_dev: [B,H,W,3]. */
int vfi_rife_interpolate(vfi_rife_t* net, int B, const int* slot0, const int* slot1,
                         const float* timestep, float* out_dev, void* stream);

/* The whole node call for a HOST clip, for hosts that do not want to re-implement the node loop:
 * frames_host [N,H,W,C] fp32 (C >= 3; alpha dropped) -> out_host [*n_out,H,W,3]: frame_0, its new frames, frame_1, ...,
 * frame_last (rife/__init__.py:225-230).  multipliers [N-1] (NULL = 2 everywhere; m <= 1 keeps the frame), skip [N-1] flags
 * (NULL = none; a skipped pair keeps the frame) as rife/__init__.py:149-174; scale_factor as the widget; `batch` tasks per
 * launch (1..32).  out_host == NULL only computes *n_out.  Synchronous; configures `net` itself; uses its own streams and
 * pinned staging.  Replaces RIFE_VFI.vfi's body, vfi_models/rife/__init__.py:149-239. */
int vfi_rife_run(vfi_rife_t* net, const float* frames_host, int N, int H, int W, int C, const int* multipliers,
                 const uint8_t* skip, float scale_factor, int batch, float* out_host, int64_t* n_out);

/* Compute units to leave free for a collective kernel that runs beside the library's launches (process-wide; 0 = none, the
 * default).  The library's persistent kernels launch one workgroup per compute unit and fill it (512 registers per SIMD, up to
 * 150 KB of LDS): a resident RCCL kernel on another stream takes whole units away and the displaced workgroups run as a second
 * round (+37 % per launch while it is resident, profiles/r04_reserved_cus.txt).  With n units reserved the persistent grids are sized
 * for the rest — results are bit-identical for any n (work items are independent of the workgroup that computes them).  The
 * multi-process path (torch.distributed: `all_gather_frames`, bench.py --gpus N) sets it around its overlapped all-gather — what the
 * reference leaves to NCCL's own scheduling.  vfi_get_reserved_cus returns the current value. */
int vfi_set_reserved_cus(int n);
int vfi_get_reserved_cus(void);

/* Host pipeline helper: hipMemcpyAsync(dst, src, bytes, kind, stream) with kind 1 = host -> device, 2 = device -> host, 3 = device ->
 * device; host memory should be pinned (pageable memory makes the call synchronous).  What the node's frame uploads / downloads
 * (`frames.to(device)` / `.cpu()`, rife/__init__.py:195-207,225-230) become: issued from worker threads without the framework's
 * per-copy dispatch (which holds the interpreter lock and queries pointer attributes: 0.6-2.7 ms per call under load, measured). */
int vfi_memcpy_async(void* dst, const void* src, int64_t bytes, int kind, void* stream);

/* A HIP stream of the caller's own (hipStreamNonBlocking, on the calling thread's current device).  The nodes' pair lanes and the HIP
 * graph captures of the op-by-op engines run on such streams: the library keys its scratch (split-K sums, attention partials, splat
 * lists) by (device, stream), and a stream drawn from a framework's shared pool can be handed to two owners.  No reference counterpart:
 * the reference runs everything on torch's current stream (rife/__init__.py:195-230). */
int vfi_stream_create(void** stream_out);
int vfi_stream_destroy(void* stream);
/* One workgroup that spins for `microseconds` of the device's real-time clock on `stream`: the probe with which a host finds out whether
 * two of its streams were bound to the same hardware queue (the HIP runtime multiplexes streams onto a handful of them — 4 by default —
 * and two streams of one queue run strictly one after the other: two pair lanes then overlap nothing).  Two spins on two streams take
 * 1x the time on different queues and 2x on one. */
int vfi_stream_spin(void* stream, int microseconds);

/* Work done by one interpolate call for roofline accounting (algorithmic, per task). */
int vfi_rife_work(vfi_rife_t* net, double* conv_flop_per_task, double* hbm_bytes_per_task);

/* ---- one process, several GPUs ------------------------------------------------------------
 * The reference's node contract is ONE vfi() call in ONE ComfyUI process (__init__.py:24-48,
 * vfi_models/rife/__init__.py:77-91), so the multi-GPU form that is a drop-in is single-process: one host thread + one
 * stream per visible device (each thread calls vfi_init(device) once), tasks block-partitioned over the devices, every
 * device copying its own shard of new frames into the shared host output tensor.  RCCL (ncclCommInitAll, bound at first
 * use by dlopen) carries the two exchanges north_star names: the weights, once, as one flat buffer, and the all-gather of new
 * frames for device-side consumers.  All vfi_comm_* calls are made by ONE thread for all devices. */

/* A second copy of `src` for the CURRENT device: same architecture and weight layout, weight arena allocated but EMPTY —
 * fill it with vfi_comm_broadcast over the arenas (vfi_rife_weights).  Configure / load_frame / interpolate as usual. */
vfi_rife_t* vfi_rife_clone_empty(const vfi_rife_t* src);
/* The network's weights as one flat device buffer (every packed tensor of vfi_rife_create, 256-byte aligned pieces). */
int vfi_rife_weights(vfi_rife_t* net, float** arena_dev, int64_t* count);

typedef struct vfi_comm vfi_comm_t;
/* ncclCommInitAll over `devices` (HIP device ids, distinct); NULL + vfi_last_error() when RCCL is unavailable. */
vfi_comm_t* vfi_comm_create(int n_devices, const int* devices);
void vfi_comm_destroy(vfi_comm_t* comm);
int vfi_comm_size(const vfi_comm_t* comm);
/* bufs_dev[i] (on devices[i], `count` floats) <- bufs_dev[root]; streams[i] = a hipStream_t created on devices[i]. */
int vfi_comm_broadcast(vfi_comm_t* comm, float* const* bufs_dev, int64_t count, int root, void* const* streams);
/* In-place all-gather-v: bufs_dev[i] holds sum(counts) floats, rank r's own block already in place at offset
 * counts[0] + ... + counts[r-1]; afterwards every buffer holds every block (grouped per-root broadcasts, so counts may differ). */
int vfi_comm_all_gather_v(vfi_comm_t* comm, float* const* bufs_dev, const int64_t* counts, void* const* streams);

/* The copy list of that in-place all-gather: `n` ranks, per-rank counts -> quadruples (source rank, destination rank, offset,
 * count) into `plan` (4 int64 per copy; cap = capacity in int64; plan may be NULL to count).  Pure host function: what the direct
 * full-mesh path (VFI_ALLGATHER=direct: one hipMemcpyPeerAsync per ordered device pair on its own stream — one xGMI link each)
 * executes.  The default is grouped per-root ncclBroadcast (VFI_ALLGATHER=rccl): the direct form has not yet run on two or more
 * physical devices.  Returns the number of copies or < 0. */
int64_t vfi_comm_plan_all_gather(int n, const int64_t* counts, int64_t* plan, int64_t cap);
/* What vfi_comm_all_gather_v uses in this process: 0 = direct peer copies, 1 = RCCL grouped broadcasts. */
int vfi_comm_all_gather_mode(void);

#ifdef __cplusplus
}
#endif
#endif /* VFI_HIP_H */
