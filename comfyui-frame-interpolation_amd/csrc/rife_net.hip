// RIFE 4.7 / 4.9 network on the device: weight residency, workspace, launch sequence.
//
// Replaces IFNet.forward (vfi_models/rife/rife_arch.py:465-732, arch "4.7") behind the C ABI of
// include/vfi_hip.h.  Per task (one new frame) the sequence is, for the 4 IFBlocks:
//   stage_in  (cat + warp + down-resize fused)           -> X
//   conv0.0, conv0.1 (3x3 stride 2, LeakyReLU)           -> A0, A1      [MFMA]
//   8 x ResConv (3x3, *beta + x, LeakyReLU)              -> A1 <-> A2   [MFMA]
//   lastconv (ConvTranspose 4x4 s2, 4 parity groups)     -> T           [MFMA]
//   flow_up (PixelShuffle + up-resize + flow/mask update) or, for the last block, final_blend.
#include <cmath>
#include <cstring>

#include "../../include/vfi_hip.h"
#include "../../include/vfi_hip_test.h"
#include "rife_ops.h"

using namespace vfi;

namespace {

struct DevBuf {
    float* p = nullptr;
    size_t n = 0;
    bool view = false;   // points into the network's weight arena (not owned)
    int ensure(size_t count) {
        if (count <= n) return 0;
        if (p && !view) (void)hipFree(p);
        p = nullptr;
        n = 0;
        view = false;
        VFI_CHECK_HIP(hipMalloc((void**)&p, count * sizeof(float)));
        n = count;
        return 0;
    }
    void release() {
        if (p && !view) (void)hipFree(p);
        p = nullptr;
        n = 0;
        view = false;
    }
};

// Weights live in ONE device allocation per network (the "arena"): vfi_rife_create stages every packed tensor into a host
// image, uploads it with one copy and points the layers' DevBufs into it.  A second device gets the same network by
// vfi_rife_clone_empty (same layout, empty arena) + one RCCL broadcast of the arena (comm.hip) — no per-tensor traffic.
struct WeightView {
    size_t delta;   // byte offset of the DevBuf inside struct vfi_rife
    size_t off, count;
};
struct Staging {
    void* net = nullptr;
    std::vector<float> host;
    std::vector<WeightView> views;
};
static thread_local Staging* g_stage = nullptr;

int upload(DevBuf& b, const std::vector<float>& h) {
    if (g_stage) {
        const size_t off = (g_stage->host.size() + 63) & ~(size_t)63;     // 256-byte aligned pieces
        g_stage->host.resize(off + h.size(), 0.f);
        std::copy(h.begin(), h.end(), g_stage->host.begin() + off);
        g_stage->views.push_back({(size_t)((char*)&b - (char*)g_stage->net), off, h.size()});
        return 0;
    }
    if (b.ensure(h.size())) return -1;
    VFI_CHECK_HIP(hipMemcpy(b.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

struct ConvLayer {
    DevBuf w, ww, bias, beta;   // ww: Winograd F(2x2,3x3) pack (3x3 stride-1 layers: the ResConvs and the 4.17 / 4.26 head; lastconv as a 3x3 layer)
    DevBuf bias3;               // lastconv as a 4 * LO-channel 3x3 layer (pack_deconv_as_conv3x3): its bias, repeated per parity group
    int Cout3_p = 0;
    int Cin = 0, Cin_p = 0, Cout = 0, Cout_p = 0;
    bool folded = false;  // residual folded into the centre tap (ResConv)
};

const int kBlockC[5] = {192, 128, 96, 64, 32};  // IFBlock widths; the fifth block exists in arch 4.26 only
constexpr int kMaxBlocks = 5;

}  // namespace

struct vfi_rife {
    ConvLayer conv00[kMaxBlocks], conv01[kMaxBlocks], res[kMaxBlocks][8], last[kMaxBlocks];
    DevBuf enc_w0, enc_b0, enc_w1, enc_b1;
    // architecture: feature planes of the frame pack (4 channels each), width of the head, its activation / mid convs
    //   4.7  : encode = Conv(3,16,s2) -> Deconv(16,4)                                   rife_arch.py:414-416
    //   4.17 : Head_417 = Conv(3,32,s2) lrelu Conv(32,32) lrelu Conv(32,32) lrelu Deconv(32,8)   :355-375
    //   4.26 : Head (16 wide, as Head_417) + 5 IFBlocks whose lastconv also returns 8 feature channels that are carried to
    //          the next block (NX), block scales [16,8,4,2,1]/scale_factor                        :378-398,451-457,267-273
    int arch = 47, NF = 1, CM = 16, CF = 4, n_mid = 0, nblocks = 4, NX = 0;
    bool enc_act = false;
    int last_out() const { return NX ? 52 : 24; }   // lastconv channels before PixelShuffle(2)
    int tplanes() const { return NX ? 4 : 2; }      // planes of the pixel-shuffled block output T
    DevBuf FEAT;                                     // [B][2][Hp][Wp][4] carried features at frame resolution
    ConvLayer enc_mid[2];
    DevBuf E2;
    int block_in(int i) const { return i == 0 ? 7 + 8 * NF : 12 + 8 * NF + NX; }   // IFBlock in_planes, rife_arch.py:410-456
    int CX(int i) const { return round_up(block_in(i), 8); }
    // geometry
    int H = 0, W = 0, Hp = 0, Wp = 0, max_batch = 0, n_slots = 0;
    int scales[kMaxBlocks] = {8, 4, 2, 1, 1};  // integer block scales (1 where the block scale is fractional)
    int up[kMaxBlocks] = {1, 1, 1, 1, 1};      // 1/scale for fractional block scales 0.5 / 0.25 (scale_factor 2 / 4)
    // workspace
    DevBuf Ppool, E, F, F2, M, X, A0, A1, A2, T;   // F2: the flow's second buffer for the fused last transition
    DevBuf X1, T1;  // frame-resolution staging around blocks that run above the frame resolution
    DevBuf Fdbg[kMaxBlocks], Xdbg[kMaxBlocks];
    bool keep = false;
    int last_B = 0;
    // weight arena
    float* arena = nullptr;
    size_t arena_n = 0;
    std::vector<WeightView> views;
    int device = 0;
    size_t pack_stride() const { return (size_t)Hp * Wp * 4 * (1 + NF); }
};

// one device allocation for all weights; every recorded DevBuf becomes a view into it
static int bind_arena(vfi_rife* net, const std::vector<WeightView>& views, size_t n, const float* host) {
    VFI_CHECK_HIP(hipGetDevice(&net->device));
    VFI_CHECK_HIP(hipMalloc((void**)&net->arena, n * sizeof(float)));
    net->arena_n = n;
    if (host) VFI_CHECK_HIP(hipMemcpy(net->arena, host, n * sizeof(float), hipMemcpyHostToDevice));
    net->views = views;
    for (const WeightView& v : views) {
        DevBuf* b = (DevBuf*)((char*)net + v.delta);
        b->p = net->arena + v.off;
        b->n = v.count;
        b->view = true;
    }
    return 0;
}

static int make_conv3x3(ConvLayer& L, const float* w, const float* b, const float* beta, int Cout, int Cin,
                        int Cin_p, bool wino = false) {
    L.Cin = Cin;
    L.Cin_p = Cin_p;
    L.Cout = Cout;
    L.Cout_p = round_up(Cout, 32);
    std::vector<float> wp, bp;
    // ResConv: lrelu((conv(x)+b)*beta + x) == lrelu((conv'(x)+b)*beta) with the identity folded into the
    // centre tap, w'[co][co][1][1] = w + 1/beta[co].  The residual then costs no memory traffic at all.
    // (x/beta*beta differs from x by <= 1 ulp.)  Only when every |beta| is comfortably away from 0.
    std::vector<float> wfold;
    L.folded = false;
    if (beta && Cin == Cout) {
        bool ok = true;
        for (int i = 0; i < Cout; ++i) ok = ok && std::fabs(beta[i]) >= 1e-2f;
        if (ok) {
            wfold.assign(w, w + (size_t)Cout * Cin * 9);
            for (int i = 0; i < Cout; ++i) wfold[((size_t)i * Cin + i) * 9 + 4] += 1.0f / beta[i];
            w = wfold.data();
            L.folded = true;
        }
    }
    pack_conv3x3(w, b, Cout, Cin, Cin_p, L.Cout_p, wp, bp);
    if (upload(L.w, wp) || upload(L.bias, bp)) return -1;
    if (wino) {
        std::vector<float> wq;
        pack_wino3x3(w, Cout, Cin, nullptr, Cin_p, L.Cout_p, wq);
        if (upload(L.ww, wq)) return -1;
    }
    if (beta) {
        std::vector<float> be(L.Cout_p, 1.f);
        for (int i = 0; i < Cout; ++i) be[i] = beta[i];
        if (upload(L.beta, be)) return -1;
    }
    return 0;
}

extern "C" {

vfi_rife_t* vfi_rife_create(int arch_ver_x10, const float* const* tensors, const int64_t* numels, int n_tensors) {
    if (arch_ver_x10 != 47 && arch_ver_x10 != 417 && arch_ver_x10 != 426) {
        set_error("vfi_rife_create: architecture code %d not implemented (47 = \"4.7\": rife47/rife49, 417 = \"4.17\": rife417, "
                  "426 = \"4.26\": rife426)", arch_ver_x10);
        return nullptr;
    }
    const int want_tensors = arch_ver_x10 == 47 ? 124 : (arch_ver_x10 == 417 ? 128 : 158);
    if (n_tensors != want_tensors) {
        set_error("vfi_rife_create: expected %d state_dict tensors for architecture code %d, got %d", want_tensors, arch_ver_x10,
                  n_tensors);
        return nullptr;
    }
    vfi_rife* net = new vfi_rife();
    net->arch = arch_ver_x10;
    Staging stage;
    stage.net = net;
    g_stage = &stage;
    if (arch_ver_x10 == 417) {
        net->NF = 2;
        net->CM = 32;
        net->CF = 8;
        net->n_mid = 2;
        net->enc_act = true;
    } else if (arch_ver_x10 == 426) {
        net->n_mid = 2;
        net->enc_act = true;
        net->nblocks = 5;
        net->NX = 8;
    }
    int k = 0;
    auto next = [&](int64_t want) -> const float* {
        if (k >= n_tensors || numels[k] != want) {
            set_error("vfi_rife_create: tensor %d has %lld elements, expected %lld", k,
                      (long long)(k < n_tensors ? numels[k] : -1), (long long)want);
            return nullptr;
        }
        return tensors[k++];
    };
    bool ok = true;
    const int LO = net->last_out(), LOp = round_up(LO, 32);
    for (int b = 0; b < net->nblocks && ok; ++b) {
        const int c = kBlockC[b], cin = net->block_in(b);
        const int cin_p = round_up(cin, 8);
        const float* w = next((int64_t)(c / 2) * cin * 9);
        const float* bi = w ? next(c / 2) : nullptr;
        ok = ok && bi && !make_conv3x3(net->conv00[b], w, bi, nullptr, c / 2, cin, cin_p);
        if (!ok) break;
        w = next((int64_t)c * (c / 2) * 9);
        bi = w ? next(c) : nullptr;
        ok = ok && bi && !make_conv3x3(net->conv01[b], w, bi, nullptr, c, c / 2, c / 2);
        for (int i = 0; i < 8 && ok; ++i) {
            const float* beta = next(c);
            w = beta ? next((int64_t)c * c * 9) : nullptr;
            bi = w ? next(c) : nullptr;
            ok = ok && bi && !make_conv3x3(net->res[b][i], w, bi, beta, c, c, c, true);
        }
        if (!ok) break;
        w = next((int64_t)c * LO * 16);
        bi = w ? next(LO) : nullptr;
        ok = ok && bi;
        if (ok) {
            ConvLayer& L = net->last[b];
            L.Cin = L.Cin_p = c;
            L.Cout = LO;
            L.Cout_p = LOp;
            std::vector<float> wp, bp;
            pack_deconv4x4(w, bi, c, LO, c, LOp, wp, bp);
            ok = !upload(L.w, wp) && !upload(L.bias, bp);
            // the same layer as ONE 3x3 convolution with 4 * LO channels on the Winograd kernel (no padding of 24 to a 32-wide N tile)
            std::vector<float> w3, b3, wq;
            pack_deconv_as_conv3x3(w, bi, c, LO, w3, b3);
            L.Cout3_p = round_up(4 * LO, 32);
            pack_wino3x3(w3.data(), 4 * LO, c, nullptr, c, L.Cout3_p, wq);
            b3.resize(L.Cout3_p, 0.f);
            ok = ok && !upload(L.ww, wq) && !upload(L.bias3, b3);
        }
    }
    if (ok) {
        const int CM = net->CM, CF = net->CF;
        const float* w0 = next((int64_t)CM * 3 * 9);
        const float* b0 = w0 ? next(CM) : nullptr;
        ok = b0 != nullptr;
        for (int m = 0; m < net->n_mid && ok; ++m) {   // Head.cnn1 / cnn2: 3x3, LeakyReLU(0.2) applied at launch
            const float* w = next((int64_t)CM * CM * 9);
            const float* bi = w ? next(CM) : nullptr;
            ok = bi && !make_conv3x3(net->enc_mid[m], w, bi, nullptr, CM, CM, CM, true);
        }
        const float* w1 = ok ? next((int64_t)CM * CF * 16) : nullptr;
        const float* b1 = w1 ? next(CF) : nullptr;
        ok = ok && b1 != nullptr;
        if (ok) {
            std::vector<float> p0((size_t)9 * 3 * CM), p1((size_t)16 * CM * CF);
            for (int co = 0; co < CM; ++co)
                for (int ci = 0; ci < 3; ++ci)
                    for (int t = 0; t < 9; ++t) p0[((size_t)t * 3 + ci) * CM + co] = w0[(co * 3 + ci) * 9 + t];
            for (int ci = 0; ci < CM; ++ci)
                for (int co = 0; co < CF; ++co)
                    for (int t = 0; t < 16; ++t) p1[((size_t)t * CM + ci) * CF + co] = w1[(ci * CF + co) * 16 + t];
            std::vector<float> vb0(b0, b0 + CM), vb1(b1, b1 + CF);
            ok = !upload(net->enc_w0, p0) && !upload(net->enc_b0, vb0) && !upload(net->enc_w1, p1) &&
                 !upload(net->enc_b1, vb1);
        }
    }
    g_stage = nullptr;
    if (ok) ok = !bind_arena(net, stage.views, stage.host.size(), stage.host.data());
    if (!ok) {
        vfi_rife_destroy(net);
        return nullptr;
    }
    return net;
}

vfi_rife_t* vfi_rife_clone_empty(const vfi_rife_t* src) {
    if (!src || !src->arena) {
        set_error("vfi_rife_clone_empty: null / unfinished source network");
        return nullptr;
    }
    vfi_rife* net = new vfi_rife();
    net->arch = src->arch, net->NF = src->NF, net->CM = src->CM, net->CF = src->CF, net->n_mid = src->n_mid;
    net->nblocks = src->nblocks, net->NX = src->NX, net->enc_act = src->enc_act;
    auto meta = [](ConvLayer& d, const ConvLayer& s_) {
        d.Cin = s_.Cin, d.Cin_p = s_.Cin_p, d.Cout = s_.Cout, d.Cout_p = s_.Cout_p, d.folded = s_.folded, d.Cout3_p = s_.Cout3_p;
    };
    for (int b = 0; b < kMaxBlocks; ++b) {
        meta(net->conv00[b], src->conv00[b]);
        meta(net->conv01[b], src->conv01[b]);
        meta(net->last[b], src->last[b]);
        for (int i = 0; i < 8; ++i) meta(net->res[b][i], src->res[b][i]);
    }
    for (int m = 0; m < 2; ++m) meta(net->enc_mid[m], src->enc_mid[m]);
    if (bind_arena(net, src->views, src->arena_n, nullptr)) {     // same layout on the CURRENT device; contents by broadcast
        vfi_rife_destroy(net);
        return nullptr;
    }
    return net;
}

int vfi_rife_weights(vfi_rife_t* net, float** arena_dev, int64_t* count) {
    VFI_REQUIRE(net && net->arena && arena_dev && count, "vfi_rife_weights: null argument / no weights");
    *arena_dev = net->arena;
    *count = (int64_t)net->arena_n;
    return 0;
}

void vfi_rife_destroy(vfi_rife_t* net) {
    if (!net) return;
    for (int b = 0; b < kMaxBlocks; ++b) {
        for (ConvLayer* L : {&net->conv00[b], &net->conv01[b], &net->last[b]}) {
            L->w.release();
            L->ww.release();
            L->bias.release();
            L->beta.release();
        }
        for (int i = 0; i < 8; ++i) {
            net->res[b][i].w.release();
            net->res[b][i].ww.release();
            net->res[b][i].bias.release();
            net->res[b][i].beta.release();
        }
        net->Fdbg[b].release();
        net->Xdbg[b].release();
    }
    for (ConvLayer& L : net->enc_mid) {
        L.w.release();
        L.ww.release();
        L.bias.release();
    }
    net->E2.release();
    for (DevBuf* d : {&net->enc_w0, &net->enc_b0, &net->enc_w1, &net->enc_b1, &net->Ppool, &net->E, &net->F, &net->F2, &net->M,
                      &net->X, &net->A0, &net->A1, &net->A2, &net->T, &net->X1, &net->T1, &net->FEAT})
        d->release();
    if (net->arena) (void)hipFree(net->arena);
    delete net;
}

int vfi_rife_configure(vfi_rife_t* net, int H, int W, int max_batch, int n_slots, float scale_factor) {
    VFI_REQUIRE(net, "vfi_rife_configure: null handle");
    VFI_REQUIRE(H > 0 && W > 0 && max_batch >= 1 && max_batch <= kMaxTasks && n_slots >= 2,
                "vfi_rife_configure: bad arguments H=%d W=%d max_batch=%d (1..%d) n_slots=%d", H, W, max_batch,
                kMaxTasks, n_slots);
    // scale_list = [8,4,2,1] / scale_factor (rife/__init__.py:157-160).  Block scales >= 1 must be 1 or even integers
    // (the down-resize then is the centre-2x2 mean); block scales 0.5 / 0.25 run the block above the frame resolution.
    // arch 4.26: [16,8,4,2,1] / scale_factor (rife/__init__.py:155-156)
    const int NB = net->nblocks;
    const float base4[kMaxBlocks] = {8.f, 4.f, 2.f, 1.f, 1.f}, base5[kMaxBlocks] = {16.f, 8.f, 4.f, 2.f, 1.f};
    const float* base = NB == 5 ? base5 : base4;
    int sc[kMaxBlocks] = {1, 1, 1, 1, 1}, up[kMaxBlocks] = {1, 1, 1, 1, 1};
    for (int i = 0; i < NB; ++i) {
        const float s = base[i] / scale_factor;
        if (s >= 1.f) {
            sc[i] = (int)s;
            up[i] = 1;
            VFI_REQUIRE((float)sc[i] == s && (sc[i] == 1 || sc[i] % 2 == 0),
                        "vfi_rife_configure: scale_factor %g gives block scale %g (supported scale_factor: 0.25, 0.5, 1, 2, 4)",
                        scale_factor, s);
        } else {
            sc[i] = 1;
            up[i] = (int)(1.f / s);
            VFI_REQUIRE((up[i] == 2 || up[i] == 4) && 1.f / (float)up[i] == s && i > 0,
                        "vfi_rife_configure: scale_factor %g gives block scale %g (supported scale_factor: 0.25, 0.5, 1, 2, 4)",
                        scale_factor, s);
        }
    }
    const int Hp = round_up(H, 64), Wp = round_up(W, 64);
    for (int i = 0; i < NB; ++i) {
        VFI_REQUIRE(Hp % (4 * sc[i]) == 0 && Wp % (4 * sc[i]) == 0,
                    "vfi_rife_configure: padded size %dx%d not divisible by 4*scale %d (the reference fails here too, "
                    "SURVEY.md App. C6)", Hp, Wp, sc[i]);
        // the block input of an up-scaled block is addressed with 32-bit element offsets
        VFI_REQUIRE((double)Hp * Wp * up[i] * up[i] * net->CX(i) < 2147483647.0,
                    "vfi_rife_configure: %dx%d is too large for scale_factor %g (block input above 2^31 elements)", H, W,
                    scale_factor);
    }
    net->H = H;
    net->W = W;
    net->Hp = Hp;
    net->Wp = Wp;
    net->max_batch = max_batch;
    net->n_slots = n_slots;
    memcpy(net->scales, sc, sizeof(sc));
    memcpy(net->up, up, sizeof(up));
    const size_t full = (size_t)Hp * Wp, B = max_batch;
    size_t x = 0, a0 = 0, a1 = 0, t = 0;
    bool any_up = false;
    for (int i = 0; i < NB; ++i) {
        any_up = any_up || up[i] > 1;
        const size_t px = full / ((size_t)sc[i] * sc[i]) * ((size_t)up[i] * up[i]);
        const size_t cx = net->CX(i);
        x = std::max(x, px * cx);
        a0 = std::max(a0, px / 4 * (kBlockC[i] / 2));
        a1 = std::max(a1, px / 16 * kBlockC[i]);
        t = std::max(t, px * 4 * net->tplanes());
    }
    // T plane 1 holds only the mask (+1 unused channel): components 2,3 are never written; keep them defined
    if (net->Ppool.ensure(net->pack_stride() * n_slots) || net->E.ensure(full / 4 * net->CM) ||
        (net->n_mid && net->E2.ensure(full / 4 * net->CM)) || net->F.ensure(B * full * 4) || net->F2.ensure(B * full * 4) ||
        net->M.ensure(B * full) || net->X.ensure(B * x) || net->A0.ensure(B * a0) || net->A1.ensure(B * a1) ||
        net->A2.ensure(B * a1) || net->T.ensure(B * t))
        return -1;
    if (any_up && (net->X1.ensure(B * full * net->CX(1)) || net->T1.ensure(B * full * 4 * net->tplanes()))) return -1;
    if (net->NX && net->FEAT.ensure(B * full * 8)) return -1;
    return 0;
}

static void fill_args(ConvArgs& a, const ConvLayer& L, const float* in, int in_cs, float* out, int out_cs, int N,
                      int Hin, int Win, int stride) {
    memset(&a, 0, sizeof(a));
    a.in = in;
    a.w = L.w.p;
    a.bias = L.bias.p;
    a.out = out;
    a.N = N;
    a.Hin = Hin;
    a.Win = Win;
    a.in_cs = in_cs;
    a.Hout = Hin / stride;
    a.Wout = Win / stride;
    a.out_cs = out_cs;
    a.Cin_p = L.Cin_p;
    a.Cout_p = L.Cout_p;
    a.Cout = L.Cout;
}

static int load_frame_impl(vfi_rife_t* net, int slot, const float* f32, const unsigned char* u8, int C, void* stream) {
    VFI_REQUIRE(net && net->Hp > 0, "vfi_rife_load_frame: network not configured");
    VFI_REQUIRE(slot >= 0 && slot < net->n_slots && C >= 3, "vfi_rife_load_frame: bad slot %d / channels %d", slot, C);
    hipStream_t st = (hipStream_t)stream;
    float* P = net->Ppool.p + (size_t)slot * net->pack_stride();
    const int Hp = net->Hp, Wp = net->Wp;
    // arch 4.7 (encode = Conv(3,16,s2) -> Deconv(16,4), no activation, no mid convs): the whole pack in one launch, the half-resolution
    // tensor E never reaches HBM.  Option fuse_encode = 0 keeps the three kernels (A/B measurements, the bit-identity test).
    const bool fuse_encode = option(kOptFuseEncode) != 0;
    if (fuse_encode && net->n_mid == 0 && net->CM == 16 && net->CF == 4 && !net->enc_act && net->NF == 1)
        return encode47_fused_launch(f32, u8, P, net->enc_w0.p, net->enc_b0.p, net->enc_w1.p, net->enc_b1.p, net->H, net->W, C, Hp, Wp, st);
    if (u8 ? prep_frame_u8_launch(u8, P, net->H, net->W, C, Hp, Wp, st) : prep_frame_launch(f32, P, net->H, net->W, C, Hp, Wp, st))
        return -1;
    if (encode_conv_launch(P, net->E.p, net->enc_w0.p, net->enc_b0.p, net->CM, net->enc_act, Hp, Wp, st)) return -1;
    float* cur = net->E.p;
    float* nxt = net->E2.p;
    for (int m = 0; m < net->n_mid; ++m) {
        ConvArgs a;
        fill_args(a, net->enc_mid[m], cur, net->CM, nxt, net->CM, 1, Hp / 2, Wp / 2, 1);
        conv3x3_taps(a);
        a.act = 1;
        a.slope = 0.2f;
        if (conv_wino_mode(-1) != 1 && net->enc_mid[m].ww.p) {
            a.w = net->enc_mid[m].ww.p;
            if (conv_wino_launch(a, 0, st, "encode_mid")) return -1;
        } else if (conv_launch(a, 1, false, -1, st, "encode_mid")) {
            return -1;
        }
        std::swap(cur, nxt);
    }
    return encode_deconv_launch(cur, P, net->enc_w1.p, net->enc_b1.p, net->CM, net->CF, Hp, Wp, st);
}

int vfi_rife_load_frame(vfi_rife_t* net, int slot, const float* frame_dev, int C, void* stream) {
    VFI_REQUIRE(frame_dev, "vfi_rife_load_frame: null frame");
    return load_frame_impl(net, slot, frame_dev, nullptr, C, stream);
}

int vfi_rife_load_frame_u8(vfi_rife_t* net, int slot, const uint8_t* frame_dev, int C, void* stream) {
    VFI_REQUIRE(frame_dev, "vfi_rife_load_frame_u8: null frame");
    return load_frame_impl(net, slot, nullptr, frame_dev, C, stream);
}

// A batch of frames: one launch for all of them where the architecture has the fused frame pack (arch 4.7: persistent workgroups,
// the next tile's source prefetched under the current tile's arithmetic), frame by frame otherwise.  Bit-identical to n single calls.
int vfi_rife_load_frames(vfi_rife_t* net, int n, const int* slots, const void* const* frames_dev, int C, int is_u8, void* stream) {
    VFI_REQUIRE(net && net->Hp > 0, "vfi_rife_load_frames: network not configured");
    VFI_REQUIRE(n >= 0 && (n == 0 || (slots && frames_dev)) && C >= 3, "vfi_rife_load_frames: bad arguments (n=%d C=%d)", n, C);
    for (int i = 0; i < n; ++i) {
        VFI_REQUIRE(slots[i] >= 0 && slots[i] < net->n_slots && frames_dev[i], "vfi_rife_load_frames: bad slot %d / null frame at %d", slots[i], i);
        for (int j = 0; j < i; ++j) VFI_REQUIRE(slots[j] != slots[i], "vfi_rife_load_frames: slot %d listed twice", slots[i]);
    }
    const bool fused = option(kOptFuseEncode) != 0 && option(kOptEncodeBatched) != 0 && net->n_mid == 0 && net->CM == 16 && net->CF == 4 &&
                       !net->enc_act && net->NF == 1;
    if (!fused || n < 2) {
        for (int i = 0; i < n; ++i)
            if (int rc = load_frame_impl(net, slots[i], is_u8 ? nullptr : (const float*)frames_dev[i], is_u8 ? (const unsigned char*)frames_dev[i] : nullptr, C, stream))
                return rc;
        return 0;
    }
    std::vector<float*> packs(n);
    for (int i = 0; i < n; ++i) packs[i] = net->Ppool.p + (size_t)slots[i] * net->pack_stride();
    return encode47_batch_launch(n, frames_dev, is_u8 != 0, packs.data(), net->enc_w0.p, net->enc_b0.p, net->enc_w1.p, net->enc_b1.p, net->H, net->W, C,
                                 net->Hp, net->Wp, (hipStream_t)stream);
}

int vfi_f32_to_u8(const float* in_dev, uint8_t* out_dev, int64_t n, void* stream) {
    VFI_REQUIRE(in_dev && out_dev && n >= 0 && ((uintptr_t)in_dev & 15) == 0 && ((uintptr_t)out_dev & 3) == 0,
                "vfi_f32_to_u8: bad arguments (in 16-byte, out 4-byte aligned)");
    return n ? f32_to_u8_launch(in_dev, out_dev, (long)n, (hipStream_t)stream) : 0;
}

int vfi_rife_interpolate(vfi_rife_t* net, int B, const int* slot0, const int* slot1, const float* timestep,
                         float* out_dev, void* stream) {
    VFI_REQUIRE(net && net->Hp > 0, "vfi_rife_interpolate: network not configured");
    VFI_REQUIRE(B >= 1 && B <= net->max_batch, "vfi_rife_interpolate: batch %d outside 1..%d", B, net->max_batch);
    hipStream_t st = (hipStream_t)stream;
    RifeTasks tasks;
    memset(&tasks, 0, sizeof(tasks));
    for (int b = 0; b < B; ++b) {
        VFI_REQUIRE(slot0[b] >= 0 && slot0[b] < net->n_slots && slot1[b] >= 0 && slot1[b] < net->n_slots,
                    "vfi_rife_interpolate: task %d uses slots %d,%d outside 0..%d", b, slot0[b], slot1[b],
                    net->n_slots - 1);
        tasks.slot0[b] = slot0[b];
        tasks.slot1[b] = slot1[b];
        tasks.t[b] = timestep[b];
    }
    const int Hp = net->Hp, Wp = net->Wp;
    static const char* kResName[kMaxBlocks] = {"resconv_c192", "resconv_c128", "resconv_c96", "resconv_c64", "resconv_c32"};
    static const char* kC00Name[kMaxBlocks] = {"conv0a_b0", "conv0a_b1", "conv0a_b2", "conv0a_b3", "conv0a_b4"};
    static const char* kC01Name[kMaxBlocks] = {"conv0b_b0", "conv0b_b1", "conv0b_b2", "conv0b_b3", "conv0b_b4"};
    static const char* kLastName[kMaxBlocks] = {"lastconv_b0", "lastconv_b1", "lastconv_b2", "lastconv_b3", "lastconv_b4"};
    const int NB = net->nblocks, TP = net->tplanes();
    const float* feat = net->NX ? net->FEAT.p : nullptr;
    bool fused_prev = false;
    // The last transition of the standard scale list (block scales 2 -> 1) can run fused into the next block's conv0.0
    // (trans1_conv0a_launch: X never goes to HBM; the flow then lives in the other of two buffers).  Option fuse0a = 0 turns it
    // off (A/B measurements); the debug taps need X and keep the un-fused path.
    const bool fuse0a_enabled = option(kOptFuse0a) != 0;
    float* Fcur = net->F.p;
    float* Falt = net->F2.p;
    bool a0_ready = false;   // conv0.0 of this block has already been computed by the fused transition
    for (int i = 0; i < NB; ++i) {
        const int s = net->scales[i], u = net->up[i];
        const int Hs = Hp / s * u, Ws = Wp / s * u;
        const int c = kBlockC[i];
        const int CX = net->CX(i), NF = net->NF;
        // X of this block: block 0 has no flow yet; later blocks get X from the fused transition kernel of the
        // previous iteration when the scale list allows it (standard [8,4,2,1]), else from stage_in.
        const bool x_ready = i > 0 && (fused_prev || a0_ready);
        if (!x_ready &&
            stage_in_launch(net->Ppool.p, net->pack_stride(), tasks, B, Fcur, net->M.p, i > 0 ? feat : nullptr,
                            u > 1 ? net->X1.p : net->X.p, Hp, Wp, s, CX, NF, i > 0, st))
            return -1;
        if (u > 1 && planar4_up_launch(net->X1.p, net->X.p, B, Hp, Wp, u, CX, /*flow plane*/ 2 + 2 * NF + net->NX / 4, st)) return -1;
        if (net->keep) {
            const size_t n = (size_t)B * Hs * Ws * CX;
            if (net->Xdbg[i].ensure(n)) return -1;
            VFI_CHECK_HIP(hipMemcpyAsync(net->Xdbg[i].p, net->X.p, n * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        ConvArgs a;
        // conv0.0: 3x3 stride 2 + LeakyReLU(0.2)
        if (!a0_ready) {
            fill_args(a, net->conv00[i], net->X.p, CX, net->A0.p, c / 2, B, Hs, Ws, 2);
            a.in_plane = Hs * Ws * 4;  // X is planar4
            conv3x3_taps(a);
            a.act = 1;
            a.slope = 0.2f;
            if (conv_launch(a, 2, false, -1, st, kC00Name[i])) return -1;
        }
        a0_ready = false;
        // conv0.1
        fill_args(a, net->conv01[i], net->A0.p, c / 2, net->A1.p, c, B, Hs / 2, Ws / 2, 2);
        conv3x3_taps(a);
        a.act = 1;
        a.slope = 0.2f;
        if (conv_launch(a, 2, false, -1, st, kC01Name[i])) return -1;
        // 8 x ResConv: lrelu(conv(x)*beta + x)
        float* cur = net->A1.p;
        float* nxt = net->A2.p;
        for (int r = 0; r < 8; ++r) {
            fill_args(a, net->res[i][r], cur, c, nxt, c, B, Hs / 4, Ws / 4, 1);
            conv3x3_taps(a);
            a.beta = net->res[i][r].beta.p;
            a.res = net->res[i][r].folded ? nullptr : cur;
            a.res_cs = c;
            a.act = 1;
            a.slope = 0.2f;
            // Winograd F(2x2,3x3) form (conv_wino.hip) unless switched off: chosen per PROCESS, never per launch, so a frame's
            // result does not depend on how it was batched
            if (conv_wino_mode(-1) != 1 && net->res[i][r].ww.p) {
                a.w = net->res[i][r].ww.p;
                if (conv_wino_launch(a, 0, st, kResName[i])) return -1;
            } else if (conv_launch(a, 1, false, -1, st, kResName[i])) {
                return -1;
            }
            std::swap(cur, nxt);
        }
        // lastconv: ConvTranspose2d(c, 24 | 52, 4, 2, 1) as 4 parity groups
        fill_args(a, net->last[i], cur, c, net->T.p, 128, B, Hs / 4, Ws / 4, 1);
        a.out_mode = 1;  // PixelShuffle(2) resolved by the epilogue: T is planar4 [B][TP][Hs][Ws][4]
        a.out_planes = TP;
        if (option(kOptDeconvWino) && conv_wino_mode(-1) != 1 && net->last[i].ww.p && c % 8 == 0) {
            // as a 3x3 stride-1 layer with 4 * LO output channels on the Winograd kernel (conv_wino.hip: pack_deconv_as_conv3x3, SHUF epilogue)
            conv3x3_taps(a);
            a.w = net->last[i].ww.p;
            a.bias = net->last[i].bias3.p;
            a.Cout = 4 * net->last[i].Cout;
            a.Cout_p = net->last[i].Cout3_p;
            if (conv_wino_launch(a, 8, st, kLastName[i])) return -1;
        } else {
            deconv4x4_taps(a);
            if (conv_launch(a, 1, true, -1, st, kLastName[i])) return -1;
        }
        const float* Tsrc = net->T.p;
        if (u > 1) {  // interpolate(tmp, scale) and flow * scale: back to the frame resolution, then as a scale-1 block
            if (t_down_launch(net->T.p, net->T1.p, B, Hp, Wp, u, TP, st)) return -1;
            Tsrc = net->T1.p;
        }
        if (i < NB - 1) {
            const int sn = net->scales[i + 1];
            const bool on_grid = u == 1 && net->up[i + 1] == 1 && s == 2 * sn;
            fused_prev = on_grid && (sn == 4 || sn == 2 || sn == 1 || (net->NX && sn == 8));
            const bool fuse0a = fuse0a_enabled && fused_prev && sn == 1 && net->NF == 1 && !net->NX && !net->keep && i > 0 &&
                                net->CX(i + 1) == 24 && kBlockC[i + 1] == 64 && net->conv00[i + 1].Cout_p == 32;
            if (fuse0a) {
                if (trans1_conv0a_launch(net->Ppool.p, net->pack_stride(), tasks, B, Tsrc, Fcur, Falt, net->conv00[i + 1].w.p,
                                         net->conv00[i + 1].bias.p, net->A0.p, Hp, Wp, 0.2f, st))
                    return -1;
                std::swap(Fcur, Falt);
                fused_prev = false;
                a0_ready = true;
            } else if (fused_prev) {
                if (net->NX ? stage_trans_x_launch(net->Ppool.p, net->pack_stride(), tasks, B, Tsrc, Fcur, net->X.p, Hp, Wp, s, sn,
                                                   i > 0, st)
                            : stage_trans_launch(net->Ppool.p, net->pack_stride(), tasks, B, Tsrc, Fcur, net->X.p, Hp, Wp,
                                                 s, sn, NF, i > 0, st))
                    return -1;
            } else {
                if (flow_up_launch(Tsrc, Fcur, net->M.p, B, Hp, Wp, s, TP, i > 0, st)) return -1;
                if (net->NX && feat_up_launch(Tsrc, net->FEAT.p, B, Hp, Wp, s, st)) return -1;
            }
            if (net->keep) {
                const size_t n = (size_t)B * Hp * Wp * 4;
                if (net->Fdbg[i].ensure(n)) return -1;
                VFI_CHECK_HIP(hipMemcpyAsync(net->Fdbg[i].p, Fcur, n * sizeof(float), hipMemcpyDeviceToDevice, st));
            }
        } else {
            float* fd = nullptr;
            if (net->keep) {
                const size_t n = (size_t)B * Hp * Wp * 4;
                if (net->Fdbg[i].ensure(n)) return -1;
                VFI_CHECK_HIP(hipMemsetAsync(net->Fdbg[i].p, 0, n * sizeof(float), st));
                fd = net->Fdbg[i].p;
            }
            if (final_blend_launch(net->Ppool.p, net->pack_stride(), tasks, B, Tsrc, Fcur, out_dev, fd, net->H,
                                   net->W, Hp, Wp, s, TP, st))
                return -1;
        }
    }
    net->last_B = B;
    return 0;
}

#ifdef VFI_TEST_TAPS      // include/vfi_hip_test.h: only in libvfi_hip_test.so
int vfi_rife_debug_keep(vfi_rife_t* net, int on) {
    VFI_REQUIRE(net, "null handle");
    net->keep = on != 0;
    return 0;
}

int64_t vfi_rife_debug_read(vfi_rife_t* net, int what, int stage, float* host_buf, int64_t cap) {
    if (!net) {
        set_error("null handle");
        return -1;
    }
    const float* src = nullptr;
    size_t n = 0;
    if (what == 0 && stage >= 0 && stage < net->nblocks) {
        src = net->Fdbg[stage].p;
        n = (size_t)net->last_B * net->Hp * net->Wp * 4;
    } else if (what == 1 && stage >= 0 && stage < net->nblocks) {
        const int s = net->scales[stage], u = net->up[stage];
        src = net->Xdbg[stage].p;
        n = (size_t)net->last_B * (net->Hp / s * u) * (net->Wp / s * u) * net->CX(stage);
    } else if (what == 2 && stage >= 0 && stage < net->n_slots) {
        src = net->Ppool.p + (size_t)stage * net->pack_stride();
        n = net->pack_stride();
    }
    if (!src || (int64_t)n > cap) {
        set_error("vfi_rife_debug_read: nothing kept for what=%d stage=%d (or buffer too small: need %lld)", what, stage,
                  (long long)n);
        return -1;
    }
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_buf, src, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("vfi_rife_debug_read: copy failed");
        return -1;
    }
    return (int64_t)n;
}
#endif  // VFI_TEST_TAPS

int vfi_rife_work(vfi_rife_t* net, double* conv_flop_per_task, double* hbm_bytes_per_task) {
    VFI_REQUIRE(net && net->Hp > 0, "vfi_rife_work: network not configured");
    const double full = (double)net->Hp * net->Wp;
    double mac = 0;
    for (int i = 0; i < net->nblocks; ++i) {
        const double px = full / ((double)net->scales[i] * net->scales[i]) * ((double)net->up[i] * net->up[i]);
        const double c = kBlockC[i];
        mac += px / 4 * (c / 2) * net->block_in(i) * 9;   // conv0.0
        mac += px / 16 * c * (c / 2) * 9;            // conv0.1
        mac += 8 * px / 16 * c * c * 9;              // ResConv x8
        mac += px / 16 * c * net->last_out() * 16;   // ConvTranspose2d: in_numel * Cout * k*k
    }
    // encode, both frames of the pair (the reference recomputes it per task; rife_arch.py:501-503)
    mac += 2 * (full / 4 * net->CM * 3 * 9 + net->n_mid * full / 4 * net->CM * net->CM * 9 + full / 4 * net->CM * net->CF * 16);
    if (conv_flop_per_task) *conv_flop_per_task = 2 * mac;
    if (hbm_bytes_per_task) {
        // 14 warps (8 of C=3, 6 of C=4): (2C+2)*4 B per pixel; 11 resizes in+out
        const double warps = (8 * (2 * 3 + 2) + 6 * (2 * 4 + 2)) * 4.0 * full;
        *hbm_bytes_per_task = warps;
    }
    return 0;
}

}  // extern "C"
