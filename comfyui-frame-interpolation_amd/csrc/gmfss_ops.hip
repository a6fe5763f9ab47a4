// GMFSS Fortuna (union) kernels — SURVEY.md 8f rank 3: everything of vfi_models/gmfss_fortuna/GMFSS_Fortuna_union_arch.py
// that is not a convolution (those run on the fp32-MFMA layer objects of gen_ops.hip): instance / layer norm, GELU,
// windowed + global attention products, global / local matching, flow propagation, convex up-sampling, MetricNet input
// assembly, the exp-metric wrapper around the summation splat, pixel shuffle, clamp + crop.  The per-element bodies live
// in gmfss_bodies.h; this file only launches them.  Built a second time with -DVFI_HOSTCHECK (tests/hostcheck) the same
// entry points run the same bodies element by element on the host, which is how the CPU test suite checks them.
#include "../../include/vfi_hip.h"
#include "gmfss_bodies.h"
#include "body_launch.h"
#ifndef VFI_HOSTCHECK
#include "gmfss_fast.h"
#endif

using namespace vfi;
using namespace vfi_gmfss;


extern "C" {

int vfi_pad_rgb(const float* frame_dev, int C, int H, int W, float* out_dev, int out_cs, int Hp, int Wp, void* stream) {
    VFI_REQUIRE(frame_dev && out_dev && C >= 3 && H > 0 && W > 0 && Hp >= H && Wp >= W && out_cs >= 3, "vfi_pad_rgb: bad arguments");
    PadRgbArgs a{frame_dev, C, H, W, out_dev, out_cs, Hp, Wp};
    return run<PadRgbArgs, pad_rgb_body>(a, (long)Hp * Wp, stream, "pad_rgb");
}

int vfi_normalize_channels(const float* in_dev, int in_cs, float* out_dev, int out_cs, int C, int64_t pixels, const float* mean_host,
                           const float* std_host, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && mean_host && std_host && C > 0 && C <= 8 && in_cs >= C && out_cs >= C && pixels > 0,
                "vfi_normalize_channels: bad arguments");
    NormChanArgs a{in_dev, in_cs, out_dev, out_cs, C, (long)pixels, {}, {}};
    for (int c = 0; c < C; ++c) {
        a.mean[c] = mean_host[c];
        a.std[c] = std_host[c];
    }
    return run<NormChanArgs, norm_chan_body>(a, (long)pixels * C, stream, "normalize_channels");
}

int vfi_prelu_scalar(const float* in_dev, int in_cs, float* out_dev, int out_cs, int C, int64_t pixels, float slope, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && C > 0 && in_cs >= C && out_cs >= C && pixels > 0, "vfi_prelu_scalar: bad arguments");
    PreluArgs a{in_dev, in_cs, out_dev, out_cs, C, (long)pixels, slope};
    return run<PreluArgs, prelu_body>(a, (long)pixels * C, stream, "prelu_scalar");
}

int vfi_instnorm_stats(const float* x_dev, int cs, int C, int N, int64_t HW, float* stats_dev, double* workspace_dev,
                       int64_t workspace_bytes, void* stream) {
    VFI_REQUIRE(x_dev && stats_dev && workspace_dev && C > 0 && cs >= C && N > 0 && HW > 0, "vfi_instnorm_stats: bad arguments");
    // strips of the first pass: as many as the workspace holds, 64 (the minimum it must hold) .. 1024
    const int64_t per_strip = (int64_t)N * C * 2 * (int64_t)sizeof(double);
    VFI_REQUIRE(workspace_bytes >= 64 * per_strip, "vfi_instnorm_stats: workspace too small");
    const int strips = (int)(workspace_bytes / per_strip < 1024 ? workspace_bytes / per_strip : 1024);
    InStatsArgs a{x_dev, cs, C, N, (long)HW, strips, workspace_dev};
    int rc;
#ifndef VFI_HOSTCHECK
    if (instnorm_partial_wg_fits(a)) rc = instnorm_partial_wg_launch(a, stream);     // a workgroup per strip (gmfss_fast.hip)
    else
#endif
        rc = run<InStatsArgs, instnorm_partial_body>(a, (long)N * strips * C, stream, "instnorm_partial");
    if (rc) return rc;
    InFinalArgs f{workspace_dev, C, N, (long)HW, strips, stats_dev, 1e-5f};
#ifndef VFI_HOSTCHECK
    return instnorm_final_wave_launch(f, stream);
#endif
    return run<InFinalArgs, instnorm_final_body>(f, (long)N * C, stream, "instnorm_final");
}

int vfi_instnorm_apply(const float* x_dev, int cs, const float* stats_dev, int C, int N, int64_t HW, int relu1, const float* add_dev,
                       int add_cs, int relu2, float* out_dev, int out_cs, void* stream) {
    VFI_REQUIRE(x_dev && stats_dev && out_dev && C > 0 && cs >= C && out_cs >= C && N > 0 && HW > 0 && (!add_dev || add_cs >= C),
                "vfi_instnorm_apply: bad arguments");
    InApplyArgs a{x_dev, cs, stats_dev, C, N, (long)HW, relu1, add_dev, add_cs, relu2, out_dev, out_cs};
    return run<InApplyArgs, instnorm_apply_body>(a, (long)N * HW * C, stream, "instnorm_apply");
}

int vfi_layernorm(const float* x_dev, int cs, int C, int64_t tokens, const float* gamma_dev, const float* beta_dev, float* out_dev,
                  int out_cs, void* stream) {
    VFI_REQUIRE(x_dev && gamma_dev && beta_dev && out_dev && C > 0 && cs >= C && out_cs >= C && tokens > 0, "vfi_layernorm: bad arguments");
    LayerNormArgs a{x_dev, cs, C, (long)tokens, gamma_dev, beta_dev, out_dev, out_cs, 1e-5f};
#ifndef VFI_HOSTCHECK
    if (layernorm_wave_fits(a)) return layernorm_wave_launch(a, stream);     // one wave per token (gmfss_fast.hip)
#endif
    return run<LayerNormArgs, layernorm_body>(a, (long)tokens, stream, "layernorm");
}

int vfi_layernorm_add(const float* x_dev, int cs, int C, int64_t tokens, const float* gamma_dev, const float* beta_dev, const float* add_dev,
                      int add_cs, float* out_dev, int out_cs, float* out2_dev, int out2_cs, void* stream) {
    VFI_REQUIRE(x_dev && gamma_dev && beta_dev && add_dev && out_dev && C > 0 && cs >= C && add_cs >= C && out_cs >= C && tokens > 0 &&
                    (!out2_dev || out2_cs >= C),
                "vfi_layernorm_add: bad arguments");
    VFI_REQUIRE(out_dev != x_dev && out2_dev != x_dev && (!out2_dev || out2_dev != add_dev), "vfi_layernorm_add: the output aliases x (or out2 aliases add)");
    LayerNormArgs a{x_dev, cs, C, (long)tokens, gamma_dev, beta_dev, out_dev, out_cs, 1e-5f, add_dev, add_cs, out2_dev, out2_cs};
#ifndef VFI_HOSTCHECK
    if (layernorm_wave_fits(a)) return layernorm_wave_launch(a, stream);
#endif
    return run<LayerNormArgs, layernorm_body>(a, (long)tokens, stream, "layernorm");
}

int vfi_gelu(float* x_dev, int cs, int C, int64_t pixels, void* stream) {
    VFI_REQUIRE(x_dev && C > 0 && cs >= C && pixels > 0, "vfi_gelu: bad arguments");
    GeluArgs a{x_dev, cs, C, (long)pixels};
    return run<GeluArgs, gelu_body>(a, (long)pixels * C, stream, "gelu");
}

int vfi_window_partition(const float* in_dev, int in_cs, float* out_dev, int out_cs, int B, int h, int w, int C, int splits, int shift_h,
                         int shift_w, int inverse, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && B > 0 && h > 0 && w > 0 && C > 0 && splits > 0 && h % splits == 0 && w % splits == 0 &&
                    in_cs >= C && out_cs >= C && shift_h >= 0 && shift_w >= 0,
                "vfi_window_partition: bad arguments (%dx%d into %d splits)", h, w, splits);
    WinPartArgs a{in_dev, in_cs, out_dev, out_cs, B, h, w, C, splits, shift_h, shift_w, inverse};
    return run<WinPartArgs, window_partition_body>(a, (long)B * h * w * C, stream, "window_partition");
}

int vfi_bmm_nt(const float* a_dev, int a_cs, const float* b_dev, int b_cs, float* out_dev, int nb, int M, int N, int K, float alpha,
               void* stream) {
    VFI_REQUIRE(a_dev && b_dev && out_dev && nb > 0 && M > 0 && N > 0 && K > 0 && a_cs >= K && b_cs >= K, "vfi_bmm_nt: bad arguments");
    BmmNtArgs a{a_dev, a_cs, b_dev, b_cs, out_dev, nb, M, N, K, alpha};
    if (K % 4 == 0 && a_cs % 4 == 0 && b_cs % 4 == 0 && ((uintptr_t)a_dev & 15) == 0 && ((uintptr_t)b_dev & 15) == 0)
        return run<BmmNtArgs, bmm_nt4_body>(a, (long)nb * ((M + 3) / 4) * ((N + 3) / 4), stream, "bmm_nt");
    return run<BmmNtArgs, bmm_nt_body>(a, (long)nb * M * N, stream, "bmm_nt");
}

int vfi_bmm_nn(const float* p_dev, const float* v_dev, int v_cs, float* out_dev, int out_cs, int nb, int M, int N, int C, void* stream) {
    VFI_REQUIRE(p_dev && v_dev && out_dev && nb > 0 && M > 0 && N > 0 && C > 0 && v_cs >= C && out_cs >= C, "vfi_bmm_nn: bad arguments");
    BmmNnArgs a{p_dev, v_dev, v_cs, out_dev, out_cs, nb, M, N, C};
    return run<BmmNnArgs, bmm_nn_body>(a, (long)nb * M * C, stream, "bmm_nn");
}

int vfi_softmax_rows(float* x_dev, int nb, int rows, int cols, const float* mask_dev, int mask_period, void* stream) {
    VFI_REQUIRE(x_dev && nb > 0 && rows > 0 && cols > 0 && (!mask_dev || mask_period > 0), "vfi_softmax_rows: bad arguments");
    SoftmaxArgs a{x_dev, nb, rows, cols, mask_dev, mask_period};
    return run<SoftmaxArgs, softmax_rows_body>(a, (long)nb * rows, stream, "softmax_rows");
}

int vfi_flow_sample(const float* in_dev, int in_cs, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N, int H, int W,
                    int C, void* stream) {
    VFI_REQUIRE(in_dev && flow_dev && out_dev && N > 0 && H > 1 && W > 1 && C > 0 && in_cs >= C && out_cs >= C && flow_cs >= 2,
                "vfi_flow_sample: bad arguments");
    FlowSampleArgs a{in_dev, in_cs, flow_dev, flow_cs, out_dev, out_cs, N, H, W, C};
    return run<FlowSampleArgs, flow_sample_body>(a, (long)N * H * W, stream, "flow_sample");
}

int vfi_resize_bilinear_ac(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int Hin, int Win, int Hout, int Wout,
                           int C, float post_mul, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C > 0 && in_cs >= C && out_cs >= C,
                "vfi_resize_bilinear_ac: bad arguments");
    ResizeAcArgs a{in_dev, in_cs, out_dev, out_cs, N, Hin, Win, Hout, Wout, C, post_mul};
    return run<ResizeAcArgs, resize_ac_body>(a, (long)N * Hout * Wout, stream, "resize_bilinear_ac");
}

int vfi_local_match(const float* f0_dev, int f0_cs, const float* f1_dev, int f1_cs, float* flow_dev, int flow_cs, int N, int H, int W,
                    int C, int radius, void* stream) {
    VFI_REQUIRE(f0_dev && f1_dev && flow_dev && N > 0 && H > 1 && W > 1 && C > 0 && f0_cs >= C && f1_cs >= C && flow_cs >= 2 &&
                    radius >= 1 && radius <= 5,
                "vfi_local_match: bad arguments");
    LocalMatchArgs a{f0_dev, f0_cs, f1_dev, f1_cs, flow_dev, flow_cs, N, H, W, C, radius};
#ifndef VFI_HOSTCHECK
    if (local_match_mfma_fits(a)) return local_match_mfma_launch(a, stream);   // GMFlow's shape: fp32-MFMA form (gmfss_fast.hip)
#endif
    return run<LocalMatchArgs, local_match_body>(a, (long)N * H * W, stream, "local_match");
}

int vfi_local_propagate(const float* q_dev, int q_cs, const float* k_dev, int k_cs, const float* flow_dev, int flow_cs, float* out_dev,
                        int out_cs, int N, int H, int W, int C, int radius, void* stream) {
    VFI_REQUIRE(q_dev && k_dev && flow_dev && out_dev && N > 0 && H > 0 && W > 0 && C > 0 && q_cs >= C && k_cs >= C && flow_cs >= 2 &&
                    out_cs >= 2 && radius >= 1 && radius <= 3 && flow_dev != out_dev,
                "vfi_local_propagate: bad arguments");
    LocalPropArgs a{q_dev, q_cs, k_dev, k_cs, flow_dev, flow_cs, out_dev, out_cs, N, H, W, C, radius};
#ifndef VFI_HOSTCHECK
    if (local_prop_coop_fits(a)) return local_prop_coop_launch(a, stream);     // GMFlow's shape: 16 lanes per pixel (gmfss_fast.hip)
#endif
    return run<LocalPropArgs, local_prop_body>(a, (long)N * H * W, stream, "local_propagate");
}

int vfi_convex_upsample(const float* mask_dev, int mask_cs, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N, int H,
                        int W, int factor, void* stream) {
    VFI_REQUIRE(mask_dev && flow_dev && out_dev && N > 0 && H > 0 && W > 0 && factor > 0 && mask_cs >= 9 * factor * factor &&
                    flow_cs >= 2 && out_cs >= 2,
                "vfi_convex_upsample: bad arguments");
    ConvexUpArgs a{mask_dev, mask_cs, flow_dev, flow_cs, out_dev, out_cs, N, H, W, factor};
    return run<ConvexUpArgs, convex_up_body>(a, (long)N * H * W * factor * factor, stream, "convex_upsample");
}

int vfi_gmfss_metric_inputs(const float* img0_dev, const float* img1_dev, int img_cs, const float* flow01_dev, const float* flow10_dev,
                            int flow_cs, float* out_dev, int out_cs, int H, int W, void* stream) {
    VFI_REQUIRE(img0_dev && img1_dev && flow01_dev && flow10_dev && out_dev && img_cs >= 3 && flow_cs >= 2 && out_cs >= 14 && H > 1 &&
                    W > 1,
                "vfi_gmfss_metric_inputs: bad arguments");
    MetricInArgs a{img0_dev, img1_dev, img_cs, flow01_dev, flow10_dev, flow_cs, out_dev, out_cs, H, W};
    return run<MetricInArgs, metric_inputs_body>(a, (long)H * W, stream, "metric_inputs");
}

int vfi_tanh_scale(float* x_dev, int cs, int C, int64_t pixels, float scale, void* stream) {
    VFI_REQUIRE(x_dev && C > 0 && cs >= C && pixels > 0, "vfi_tanh_scale: bad arguments");
    TanhArgs a{x_dev, cs, C, (long)pixels, scale};
    return run<TanhArgs, tanh_scale_body>(a, (long)pixels * C, stream, "tanh_scale");
}

int vfi_splat_prep(const float* x_dev, int x_cs, const float* z_dev, int z_cs, const float* flow_dev, int flow_cs, float* out_dev,
                   float* flow_out_dev, int C, int64_t pixels, float z_scale, float flow_scale, void* stream) {
    VFI_REQUIRE(x_dev && z_dev && flow_dev && out_dev && flow_out_dev && C > 0 && x_cs >= C && z_cs >= 1 && flow_cs >= 2 && pixels > 0,
                "vfi_splat_prep: bad arguments");
    SplatPrepArgs a{x_dev, x_cs, z_dev, z_cs, flow_dev, flow_cs, out_dev, flow_out_dev, C, (long)pixels, z_scale, flow_scale};
    return run<SplatPrepArgs, splat_prep_body>(a, (long)pixels * (C + 1), stream, "splat_prep");
}

int vfi_splat_normalize(const float* splat_dev, float* out_dev, int out_cs, int C, int64_t pixels, void* stream) {
    VFI_REQUIRE(splat_dev && out_dev && C > 0 && out_cs >= C && pixels > 0, "vfi_splat_normalize: bad arguments");
    SplatNormArgs a{splat_dev, out_dev, out_cs, C, (long)pixels};
    return run<SplatNormArgs, splat_norm_body>(a, (long)pixels * C, stream, "splat_normalize");
}

int vfi_pixel_shuffle2(const float* in_dev, int in_cs, float* out_dev, int out_cs, int N, int H, int W, int C, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && N > 0 && H > 0 && W > 0 && C > 0 && in_cs >= 4 * C && out_cs >= C, "vfi_pixel_shuffle2: bad arguments");
    PixShufArgs a{in_dev, in_cs, out_dev, out_cs, N, H, W, C};
    return run<PixShufArgs, pixel_shuffle2_body>(a, (long)N * 4 * H * W * C, stream, "pixel_shuffle2");
}

int vfi_clamp_crop(const float* in_dev, int in_cs, int Hp, int Wp, float* out_dev, int H, int W, int C, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && Hp >= H && Wp >= W && H > 0 && W > 0 && C > 0 && in_cs >= C, "vfi_clamp_crop: bad arguments");
    ClampCropArgs a{in_dev, in_cs, Hp, Wp, out_dev, H, W, C};
    return run<ClampCropArgs, clamp_crop_body>(a, (long)H * W * C, stream, "clamp_crop");
}

}  // extern "C"
