// IFUNet kernels — SURVEY.md 8f rank 4, second half: the parts of vfi_models/ifunet/IFUNet_arch.py that are not convolutions
// (those run on the fp32-MFMA layer objects): CBAM's channel and spatial gates, the 4-channel convex flow up-sampling of the
// IFBlocks, the mask blends of IFUNetModel / ResynNet.  Bodies in ifunet_bodies.h, launch scheme in body_launch.h.
// The CBAM reductions dispatch to the cooperative kernels of ifunet_fast.hip on the device; built with -DVFI_HOSTCHECK
// (tests/hostcheck) every entry point runs its body on the host.
#include "../../include/vfi_hip.h"
#include "ifunet_bodies.h"
#include "body_launch.h"
#ifndef VFI_HOSTCHECK
#include "ifunet_fast.h"
#endif

using namespace vfi;
using namespace vfi_ifunet;

extern "C" {

int vfi_channel_pool(const float* x_dev, int cs, int C, int N, int64_t HW, float* stats_dev, void* workspace_dev, int64_t workspace_bytes,
                     void* stream) {
    VFI_REQUIRE(x_dev && stats_dev && workspace_dev && C > 0 && cs >= C && N > 0 && HW > 0, "vfi_channel_pool: bad arguments");
    // strips of the first pass: as many as the workspace holds, 64 (the minimum it must hold) .. 128
    const int64_t per_strip = (int64_t)N * C * (int64_t)(sizeof(double) + sizeof(float));
    VFI_REQUIRE(workspace_bytes >= 64 * per_strip, "vfi_channel_pool: workspace too small");
    // (128 at most: the second pass reads every strip's partials of a channel, 64 of them per load instruction)
    const int strips = (int)(workspace_bytes / per_strip < 128 ? workspace_bytes / per_strip : 128);
    const int64_t cells = (int64_t)N * strips * C;
    double* psum = (double*)workspace_dev;
    float* pmax = (float*)(psum + cells);
    PoolPartArgs a{x_dev, cs, C, N, (long)HW, strips, psum, pmax};
    PoolFinalArgs f{psum, pmax, C, N, (long)HW, strips, stats_dev};
#ifndef VFI_HOSTCHECK
    if (chan_pool_partial_wg_fits(a)) {
        if (int rc = chan_pool_partial_wg_launch(a, stream)) return rc;
        return chan_pool_final_wave_launch(f, stream);
    }
#endif
    int rc = run<PoolPartArgs, chan_pool_partial_body>(a, (long)cells, stream, "channel_pool_partial");
    if (rc) return rc;
    return run<PoolFinalArgs, chan_pool_final_body>(f, (long)N * C, stream, "channel_pool_final");
}

int vfi_cbam_gate(const float* stats_dev, const float* w1_dev, const float* b1_dev, const float* w2_dev, const float* b2_dev, int C, int R,
                  int N, float* scale_dev, void* stream) {
    VFI_REQUIRE(stats_dev && w1_dev && b1_dev && w2_dev && b2_dev && scale_dev && C > 0 && R > 0 && N > 0, "vfi_cbam_gate: bad arguments");
    GateArgs a{stats_dev, w1_dev, b1_dev, w2_dev, b2_dev, C, R, N, scale_dev};
#ifndef VFI_HOSTCHECK
    if (cbam_gate_wg_fits(a)) return cbam_gate_wg_launch(a, stream);
#endif
    return run<GateArgs, cbam_gate_body>(a, (long)N * C, stream, "cbam_gate");
}

int vfi_cbam_scale_compress(const float* x_dev, int cs, const float* scale_dev, int C, int N, int64_t HW, float* xs_dev, int xs_cs,
                            float* comp_dev, void* stream) {
    VFI_REQUIRE(x_dev && scale_dev && xs_dev && comp_dev && C > 0 && cs >= C && xs_cs >= C && N > 0 && HW > 0,
                "vfi_cbam_scale_compress: bad arguments");
    ScaleCompArgs a{x_dev, cs, scale_dev, C, N, (long)HW, xs_dev, xs_cs, comp_dev};
#ifndef VFI_HOSTCHECK
    return cbam_scale_compress_wave_launch(a, stream);
#endif
    return run<ScaleCompArgs, cbam_scale_compress_body>(a, (long)N * HW, stream, "cbam_scale_compress");
}

int vfi_cbam_spatial(float* xs_dev, int cs, const float* comp_dev, const float* w_dev, float bn_a, float bn_b, int C, int N, int H, int W,
                     void* stream) {
    VFI_REQUIRE(xs_dev && comp_dev && w_dev && C > 0 && cs >= C && N > 0 && H > 0 && W > 0, "vfi_cbam_spatial: bad arguments");
    SpatialArgs a{xs_dev, cs, comp_dev, w_dev, bn_a, bn_b, C, N, H, W};
#ifndef VFI_HOSTCHECK
    return cbam_spatial_wave_launch(a, stream);
#endif
    return run<SpatialArgs, cbam_spatial_body>(a, (long)N * H * W, stream, "cbam_spatial");
}

int vfi_convex_upsample_c(const float* mask_dev, int mask_cs, const float* flow_dev, int flow_cs, float* out_dev, int out_cs, int N, int H,
                          int W, int factor, int flow_channels, void* stream) {
    VFI_REQUIRE(mask_dev && flow_dev && out_dev && N > 0 && H > 0 && W > 0 && factor > 0 && mask_cs >= 9 * factor * factor &&
                    flow_channels > 0 && flow_channels <= 8 && flow_cs >= flow_channels && out_cs >= flow_channels,
                "vfi_convex_upsample_c: bad arguments");
    ConvexUpCArgs a{mask_dev, mask_cs, flow_dev, flow_cs, out_dev, out_cs, N, H, W, factor, flow_channels};
    return run<ConvexUpCArgs, convex_up_c_body>(a, (long)N * H * W * factor * factor, stream, "convex_upsample_c");
}

int vfi_lerp_mask(const float* a_dev, int a_cs, const float* b_dev, int b_cs, const float* mask_dev, int mask_cs, float* out_dev,
                  int out_cs, int C, int64_t pixels, void* stream) {
    VFI_REQUIRE(a_dev && b_dev && mask_dev && out_dev && C > 0 && a_cs >= C && b_cs >= C && out_cs >= C && mask_cs >= 1 && pixels > 0,
                "vfi_lerp_mask: bad arguments");
    LerpArgs a{a_dev, a_cs, b_dev, b_cs, mask_dev, mask_cs, out_dev, out_cs, C, (long)pixels};
    return run<LerpArgs, lerp_mask_body>(a, (long)pixels * C, stream, "lerp_mask");
}

int vfi_add_clamp01(const float* a_dev, int a_cs, const float* b_dev, int b_cs, float* out_dev, int out_cs, int C, int64_t pixels,
                    void* stream) {
    VFI_REQUIRE(a_dev && b_dev && out_dev && C > 0 && a_cs >= C && b_cs >= C && out_cs >= C && pixels > 0, "vfi_add_clamp01: bad arguments");
    AddClampArgs a{a_dev, a_cs, b_dev, b_cs, out_dev, out_cs, C, (long)pixels};
    return run<AddClampArgs, add_clamp_body>(a, (long)pixels * C, stream, "add_clamp01");
}

int vfi_ifunet_blend(const float* img0_dev, const float* img1_dev, const float* deg_dev, int img_cs, const float* mask0_dev,
                     const float* mask1_dev, int mask_cs, float* out_dev, int Hp, int Wp, int H, int W, void* stream) {
    VFI_REQUIRE(img0_dev && img1_dev && deg_dev && mask0_dev && mask1_dev && out_dev && img_cs >= 3 && mask_cs >= 1 && Hp >= H && Wp >= W &&
                    H > 0 && W > 0,
                "vfi_ifunet_blend: bad arguments");
    ResynBlendArgs a{img0_dev, img1_dev, deg_dev, img_cs, mask0_dev, mask1_dev, mask_cs, out_dev, Hp, Wp, H, W};
    return run<ResynBlendArgs, resyn_blend_body>(a, (long)H * W, stream, "ifunet_blend");
}

int vfi_fill_channels(float* out_dev, int cs, int C, int64_t pixels, float value, void* stream) {
    VFI_REQUIRE(out_dev && C > 0 && cs >= C && pixels > 0, "vfi_fill_channels: bad arguments");
    FillArgs a{out_dev, cs, C, (long)pixels, value};
    return run<FillArgs, fill_body>(a, (long)pixels * C, stream, "fill_channels");
}

}  // extern "C"
