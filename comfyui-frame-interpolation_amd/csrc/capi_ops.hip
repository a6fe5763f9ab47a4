// Single-op C entry points (include/vfi_hip.h): thin argument checking + host-side weight packing
// around the kernels, used by the parity tests and available to other node implementations.
#include <cstdlib>
#include <cstring>

#include "../../include/vfi_hip.h"
#include "../../include/vfi_hip_test.h"
#include "rife_ops.h"

using namespace vfi;

namespace {
struct Tmp {
    float* p = nullptr;
    ~Tmp() {
        if (p) (void)hipFree(p);
    }
    int put(const std::vector<float>& h) {
        VFI_CHECK_HIP(hipMalloc((void**)&p, std::max<size_t>(h.size(), 4) * sizeof(float)));
        VFI_CHECK_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    }
    int alloc(size_t n) {
        VFI_CHECK_HIP(hipMalloc((void**)&p, std::max<size_t>(n, 4) * sizeof(float)));
        return 0;
    }
};
}  // namespace

extern "C" {

int vfi_warp_border(const float* in_dev, const float* flow_dev, float* out_dev, int N, int H, int W, int C,
                    void* stream) {
    VFI_REQUIRE(in_dev && flow_dev && out_dev && N > 0 && H > 1 && W > 1 && C > 0, "vfi_warp_border: bad arguments");
    return warp_border_launch(in_dev, flow_dev, out_dev, N, H, W, C, (hipStream_t)stream);
}

int vfi_conv3x3(const float* in_dev, const float* weight_host, const float* bias_host, const float* beta_host,
                float* out_dev, int N, int H, int W, int Cin, int Cout, int stride, int act, float slope, int variant,
                void* stream) {
    VFI_REQUIRE(in_dev && weight_host && out_dev && N > 0 && H > 0 && W > 0, "vfi_conv3x3: bad arguments");
    VFI_REQUIRE(stride == 1 || stride == 2, "vfi_conv3x3: stride %d", stride);
    VFI_REQUIRE(Cin % 4 == 0, "vfi_conv3x3: Cin=%d must be a multiple of 4 (NHWC float4 rows)", Cin);
    VFI_REQUIRE(!beta_host || (Cin == Cout && stride == 1), "vfi_conv3x3: beta/residual needs Cin==Cout, stride 1");
    hipStream_t st = (hipStream_t)stream;
    // K chunk of the variant decides the channel padding of the packed weights
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.N = N;
    a.Hin = H;
    a.Win = W;
    a.Hout = (H + 2 - 3) / stride + 1;
    a.Wout = (W + 2 - 3) / stride + 1;
    a.Cout_p = round_up(Cout, 32);
    a.Cout = Cout;
    conv3x3_taps(a);
    int v = variant;
    a.Cin_p = round_up(Cin, 8);
    const bool wino = v == 100 || v == 101;      // Winograd F(2x2,3x3) form (conv_wino.hip): 16x8 / 32x4 pixel regions
    VFI_REQUIRE(!wino || stride == 1, "vfi_conv3x3: the Winograd variants are stride 1");
    if (v < 0) v = conv_pick_variant(a, stride, false);
    VFI_REQUIRE(wino || conv_variant_lookup(v), "vfi_conv3x3: bad variant %d", v);
    const int ck = wino ? 8 : conv_variant_lookup(v)->ck;
    a.Cin_p = round_up(Cin, ck);
    // the activation tensor must physically hold Cin_p channels: copy into a padded temp if not
    Tmp inpad;
    const float* in_use = in_dev;
    int in_cs = Cin;
    if (a.Cin_p != Cin) {
        const size_t px = (size_t)N * H * W;
        if (inpad.alloc(px * a.Cin_p)) return -1;
        VFI_CHECK_HIP(hipMemsetAsync(inpad.p, 0, px * a.Cin_p * sizeof(float), st));
        VFI_CHECK_HIP(hipMemcpy2DAsync(inpad.p, a.Cin_p * sizeof(float), in_dev, Cin * sizeof(float),
                                       Cin * sizeof(float), px, hipMemcpyDeviceToDevice, st));
        in_use = inpad.p;
        in_cs = a.Cin_p;
    }
    std::vector<float> wp, bp;
    pack_conv3x3(weight_host, bias_host, Cout, Cin, a.Cin_p, a.Cout_p, wp, bp);
    if (wino) pack_wino3x3(weight_host, Cout, Cin, nullptr, a.Cin_p, a.Cout_p, wp);
    Tmp dw, db, dbeta;
    if (dw.put(wp) || db.put(bp)) return -1;
    if (beta_host) {
        std::vector<float> be(a.Cout_p, 1.f);
        for (int i = 0; i < Cout; ++i) be[i] = beta_host[i];
        if (dbeta.put(be)) return -1;
        a.beta = dbeta.p;
        a.res = in_dev;
        a.res_cs = Cin;
    }
    a.in = in_use;
    a.in_cs = in_cs;
    a.w = dw.p;
    a.bias = db.p;
    a.out = out_dev;
    a.out_cs = Cout;
    a.act = act;
    a.slope = slope;
    if (wino ? conv_wino_launch(a, v == 100 ? 8 : 16, st, "conv3x3_wino") : conv_launch(a, stride, false, v, st, nullptr)) return -1;
    VFI_CHECK_HIP(hipStreamSynchronize(st));  // temporaries are freed on return
    return 0;
}

#ifdef VFI_TEST_TAPS      // include/vfi_hip_test.h: only in libvfi_hip_test.so
int vfi_test_conv_algo(int mode) { return conv_wino_mode(mode); }
int vfi_test_wino_probe_read(uint32_t* out32) {
    VFI_REQUIRE(out32, "vfi_test_wino_probe_read: null buffer");
    return wino_probe_read(out32);
}

int64_t vfi_test_pack_wino3x3(const float* weight_host, int Cout, int Cin, const int* chan_map, int Cin_p, float* out_host, int64_t cap) {
    if (!weight_host || !out_host || Cout <= 0 || Cin <= 0 || Cin_p % 8 || Cin_p < Cin) {
        set_error("vfi_test_pack_wino3x3: bad arguments");
        return -1;
    }
    std::vector<float> wp;
    pack_wino3x3(weight_host, Cout, Cin, chan_map, Cin_p, round_up(Cout, 32), wp);
    if ((int64_t)wp.size() > cap) {
        set_error("vfi_test_pack_wino3x3: buffer too small (need %lld floats)", (long long)wp.size());
        return -1;
    }
    memcpy(out_host, wp.data(), wp.size() * sizeof(float));
    return (int64_t)wp.size();
}

int64_t vfi_test_pack_deconv3x3(const float* weight_host, const float* bias_host, int Cin, int LO, float* w3_host, float* b3_host, int64_t cap) {
    if (!weight_host || !w3_host || !b3_host || Cin <= 0 || LO <= 0) {
        set_error("vfi_test_pack_deconv3x3: bad arguments");
        return -1;
    }
    std::vector<float> w3, b3;
    pack_deconv_as_conv3x3(weight_host, bias_host, Cin, LO, w3, b3);
    if ((int64_t)w3.size() > cap) {
        set_error("vfi_test_pack_deconv3x3: buffer too small (need %lld floats)", (long long)w3.size());
        return -1;
    }
    memcpy(w3_host, w3.data(), w3.size() * sizeof(float));
    memcpy(b3_host, b3.data(), b3.size() * sizeof(float));
    return (int64_t)w3.size();
}

int vfi_conv3x3_naive(const float* in_dev, const float* weight_host, const float* bias_host, const float* beta_host,
                      float* out_dev, int N, int H, int W, int Cin, int Cout, int stride, int act, float slope,
                      void* stream) {
    VFI_REQUIRE(in_dev && weight_host && out_dev, "vfi_conv3x3_naive: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    Tmp dw, db, dbeta;
    std::vector<float> w(weight_host, weight_host + (size_t)Cout * Cin * 9);
    if (dw.put(w)) return -1;
    if (bias_host) {
        std::vector<float> b(bias_host, bias_host + Cout);
        if (db.put(b)) return -1;
    }
    if (beta_host) {
        std::vector<float> b(beta_host, beta_host + Cout);
        if (dbeta.put(b)) return -1;
    }
    if (conv_naive_launch(in_dev, dw.p, db.p, dbeta.p, out_dev, N, H, W, Cin, Cin, Cout, stride, act, slope, st))
        return -1;
    VFI_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}
#endif  // VFI_TEST_TAPS

int vfi_deconv4x4_ps2(const float* in_dev, const float* weight_host, const float* bias_host, float* out_dev, int N,
                      int H, int W, int Cin, int Cout, void* stream) {
    VFI_REQUIRE(in_dev && weight_host && out_dev, "vfi_deconv4x4_ps2: bad arguments");
    VFI_REQUIRE(Cin % 16 == 0 && Cout % 4 == 0 && Cout <= 32, "vfi_deconv4x4_ps2: Cin=%d (x16) Cout=%d (x4, <=32)", Cin,
                Cout);
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> wp, bp;
    pack_deconv4x4(weight_host, bias_host, Cin, Cout, Cin, 32, wp, bp);
    Tmp dw, db, T;
    if (dw.put(wp) || db.put(bp) || T.alloc((size_t)N * H * W * 128)) return -1;  // planar4 [N][2][4H][4W][4]
    VFI_CHECK_HIP(hipMemsetAsync(T.p, 0, (size_t)N * H * W * 128 * sizeof(float), st));
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in_dev;
    a.w = dw.p;
    a.bias = db.p;
    a.out = T.p;
    a.N = N;
    a.Hin = a.Hout = H;
    a.Win = a.Wout = W;
    a.in_cs = Cin;
    a.out_cs = 128;
    a.Cin_p = Cin;
    a.Cout_p = 32;
    a.Cout = Cout;
    a.out_mode = 1;
    Tmp dw3, db3;
    if (option(kOptDeconvWino) && conv_wino_mode(-1) != 1 && option(kOptGroupedVariant) < 0 && Cin % 8 == 0) {
        // the form the RIFE network uses: one 3x3 layer with 4 * Cout channels on the Winograd kernel, pixel-shuffle epilogue
        std::vector<float> w3, b3, wq;
        pack_deconv_as_conv3x3(weight_host, bias_host, Cin, Cout, w3, b3);
        const int cp = round_up(4 * Cout, 32);
        pack_wino3x3(w3.data(), 4 * Cout, Cin, nullptr, Cin, cp, wq);
        b3.resize(cp, 0.f);
        if (dw3.put(wq) || db3.put(b3)) return -1;
        conv3x3_taps(a);
        a.w = dw3.p;
        a.bias = db3.p;
        a.Cout = 4 * Cout;
        a.Cout_p = cp;
        if (conv_wino_launch(a, 8, st, "deconv4x4_wino")) return -1;
    } else {
        deconv4x4_taps(a);
        if (conv_launch(a, 1, true, -1, st, nullptr)) return -1;
    }
    if (t_to_nhwc_launch(T.p, out_dev, N, H, W, Cout / 4, st)) return -1;
    VFI_CHECK_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
