// M2M interpolator as a C-side object ("same triplet for M2M"): vfi_m2m_create / vfi_m2m_prepare /
// vfi_m2m_render / vfi_m2m_destroy — weights resident, workspace owned, the ~110 launches of a frame pair issued by one call.
//
// Replaces M2M_PWC.forward (vfi_models/m2m/M2M_arch.py:894-1037): replicate padding + joint normalisation :903-934, the
// bidirectional PWC flow network on half-resolution images :415-546,936-939 (Extractor, 5 Decoders with the 9x9 cost volume),
// MotionRefineNet / EncDec :606-890 (image pyramid, encoder, cube attention, decoder, 8 flow residuals + mask), the photometric
// metric and — per timestep — forwarp_mframe_mask :551-581,945-1037.
// Both directions of a pair share every weight and run as a batch of 2 (image 0 = frame0->frame1 quantities, image 1 = the
// reverse), "partner" reads use the swap flag of the kernels; torch.cat along channels is a channel-window offset.
// Everything timestep independent runs once per pair (prepare), only the splat per timestep (render) — the reference
// re-runs the whole network for every timestep (vfi_utils.py:201-211); results are identical.
// Built on the library's own entry points (vfi_conv_forward_ex on the fp32 matrix cores, vfi_costvol9x9, vfi_warp_m2m,
// vfi_resize_bilinear, vfi_avgpool2, vfi_pool_mean, vfi_m2m_*, vfi_softsplat_sum).  Round 1 issued this sequence from Python;
// the Python M2MEngine now wraps this object.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/vfi_hip.h"
#include "../../include/vfi_hip_test.h"
#include "vfi_common.h"

using namespace vfi;

namespace {

constexpr int RATIO = 4;       // M2M_PWC.forward default ratio (the node never overrides it, m2m/__init__.py:51-55)
constexpr int DEC_CS = 120;    // decoder input window: [feature 32 | cost volume 81 | flow 2 | pad] (115 -> x8)
constexpr int FLOW_OFF = 113;
constexpr int CC = 16;         // M2M_arch.py:586

struct Ten {   // [n][h][w][c] fp32, zero-initialised
    float* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
};

struct Layer {
    vfi_conv_t* h = nullptr;
    int kind = 0, stride = 1;
    float slope = 0.f;   // single-parameter PReLU that follows the layer (act 1), where there is one
};

}  // namespace

struct vfi_m2m {
    Layer ext[3][3], dec[5][6], pyr[4][2], down[4][2], up[4], head, cube[3];
    float alpha = 0.f;
    std::vector<vfi_conv_t*> all;
    // workspace
    int H = 0, W = 0, Hp = 0, Wp = 0;
    bool prepared = false;
    Ten d0, imh, decb[5], flow[5], enc[4], fl[5], s3, pc, ph, pw, cc, ch, cw, xf, r, tf, e, sin, sfl, sout, img4;
    float* tile_ranges = nullptr;      // [8][tiles][4]: per 32x32 tile the range of every refined flow field (vfi_m2m_photo_tiles)
    float* smax = nullptr;             // [8]: max |tf_s|
    float* stats = nullptr;
    void* ws = nullptr;
    // r6 A/B form (option m2m_side, off): the image-pyramid convolutions of the refinement network depend on the normalised frames only,
    // not on the flow, and can run on this side stream beside the PWC flow network (fork / join by events).  Bit-identical, and measured
    // neutral for one pair (7.00-7.04 vs 7.0-7.1 ms) and 8 % slower under three pair lanes (lanes.py already fill the coarse levels' holes)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int dech[5][2], ench[4][2];
    std::map<std::tuple<std::string, int, int, int>, Ten> scratch;
    std::vector<void*> owned;
    int64_t owned_bytes = 0;           // vfi_m2m_workspace_bytes (the tensors; a few KB of control words are not counted)
};

namespace {

int alloc_ten(vfi_m2m* m, Ten& t, int n, int h, int w, int c, hipStream_t st = nullptr) {
    t.n = n, t.h = h, t.w = w, t.c = c;
    const size_t bytes = (size_t)n * h * w * c * sizeof(float);
    VFI_CHECK_HIP(hipMalloc((void**)&t.p, bytes));
    m->owned.push_back(t.p);
    m->owned_bytes += (int64_t)bytes;
    // zero fill ordered with the forward's kernels: a NULL-stream memset is not ordered against a non-blocking side stream (torch's)
    // and could clear a lazily allocated scratch tensor AFTER its first producer ran
    VFI_CHECK_HIP(hipMemsetAsync(t.p, 0, bytes, st));
    if (!st) VFI_CHECK_HIP(hipStreamSynchronize(nullptr));
    return 0;
}

void free_workspace(vfi_m2m* m) {
    for (void* p : m->owned) (void)hipFree(p);
    m->owned.clear();
    m->owned_bytes = 0;
    m->scratch.clear();
    m->stats = nullptr;
    m->tile_ranges = m->smax = nullptr;
    m->sin = m->sfl = m->sout = Ten();
    m->ws = nullptr;
    m->H = m->W = 0;
    m->prepared = false;
}

int ensure_workspace(vfi_m2m* m, int H, int W) {
    if (m->H == H && m->W == W) return 0;
    VFI_CHECK_HIP(hipDeviceSynchronize());
    free_workspace(m);
    const int mult = RATIO * 16;
    const int Hp = (H + mult - 1) / mult * mult, Wp = (W + mult - 1) / mult * mult;
    m->Hp = Hp, m->Wp = Wp;
    const int h = Hp / 2, w = Wp / 2;
    if (alloc_ten(m, m->d0, 2, Hp, Wp, 8)) return -1;                 // [flow 2 | normalised image 3 | warped partner image 3]
    VFI_CHECK_HIP(hipMalloc((void**)&m->stats, 2 * sizeof(float)));
    m->owned.push_back(m->stats);
    VFI_CHECK_HIP(hipMemset(m->stats, 0, 2 * sizeof(float)));
    VFI_CHECK_HIP(hipMalloc(&m->ws, 16384));
    m->owned.push_back(m->ws);
    VFI_CHECK_HIP(hipMemset(m->ws, 0, 16384));
    if (alloc_ten(m, m->imh, 2, h, w, 8)) return -1;                  // half-resolution images for the flow network
    for (int l = 0; l < 5; ++l) {
        m->dech[l][0] = h >> (l + 1), m->dech[l][1] = w >> (l + 1);
        if (alloc_ten(m, m->decb[l], 2, m->dech[l][0], m->dech[l][1], DEC_CS) || alloc_ten(m, m->flow[l], 2, m->dech[l][0], m->dech[l][1], 2))
            return -1;
    }
    for (int l = 0; l < 4; ++l) {
        m->ench[l][0] = Hp >> (l + 1), m->ench[l][1] = Wp >> (l + 1);
        if (alloc_ten(m, m->enc[l], 2, m->ench[l][0], m->ench[l][1], 96 << l)) return -1;   // [s | c | warp(partner s,c)], later [s | x]
        if (alloc_ten(m, m->fl[l + 1], 2, m->ench[l][0], m->ench[l][1], 2)) return -1;      // flows at 1/2 .. 1/16
    }
    const int e3h = m->ench[3][0], e3w = m->ench[3][1];
    if (alloc_ten(m, m->s3, 2, e3h, e3w, 256) || alloc_ten(m, m->pc, 2, 1, 1, 256) || alloc_ten(m, m->ph, 2, e3h, 1, 256) ||
        alloc_ten(m, m->pw, 2, 1, e3w, 256) || alloc_ten(m, m->cc, 2, 1, 1, 4096) || alloc_ten(m, m->ch, 2, e3h, 1, 16) ||
        alloc_ten(m, m->cw, 2, 1, e3w, 16) || alloc_ten(m, m->xf, 2, Hp, Wp, 16) || alloc_ten(m, m->r, 2, Hp, Wp, 12) ||
        alloc_ten(m, m->tf, 8, Hp, Wp, 2) || alloc_ten(m, m->e, 8, Hp, Wp, 1) || alloc_ten(m, m->img4, 2, Hp, Wp, 4))
        return -1;
    {
        const size_t tiles = (size_t)((Hp + 31) / 32) * ((Wp + 31) / 32);
        VFI_CHECK_HIP(hipMalloc((void**)&m->tile_ranges, 8 * tiles * 4 * sizeof(float)));
        m->owned.push_back(m->tile_ranges);
        VFI_CHECK_HIP(hipMalloc((void**)&m->smax, 8 * sizeof(float)));
        m->owned.push_back(m->smax);
        VFI_CHECK_HIP(hipMemset(m->smax, 0, 8 * sizeof(float)));
        VFI_CHECK_HIP(hipStreamSynchronize(nullptr));      // (the fills above are not ordered against the caller's non-blocking stream)
    }
    m->H = H, m->W = W;
    return 0;
}

int tmp(vfi_m2m* m, const char* name, int h, int w, int c, Ten** out) {
    auto key = std::make_tuple(std::string(name), h, w, c);
    auto it = m->scratch.find(key);
    if (it == m->scratch.end()) {
        Ten t;
        if (alloc_ten(m, t, 2, h, w, c)) return -1;
        it = m->scratch.emplace(key, t).first;
    }
    *out = &it->second;
    return 0;
}

// out[..., doff : doff + cout] = act(layer(src[..., soff : soff + cin_phys]) (+ res))
int run(const Layer& L, const Ten& src, int soff, const Ten& dst, int doff, int act, float slope, hipStream_t st, const Ten* res = nullptr,
        int roff = 0) {
    return vfi_conv_forward_ex(L.h, src.p + soff, src.c, src.h, src.w, dst.p + doff, dst.c, src.n, act, slope, 0.f, 0.f,
                               res ? res->p + roff : nullptr, res ? res->c : 0, st);
}
int resize(const Ten& src, int soff, const Ten& dst, int doff, int c, float mul, hipStream_t st) {
    return vfi_resize_bilinear(src.p + soff, src.c, dst.p + doff, dst.c, src.n, src.h, src.w, dst.h, dst.w, c, mul, st);
}
int warp(const Ten& src, int soff, int c, const Ten& flow, int foff, const Ten& dst, int doff, hipStream_t st, int swap = 1) {
    return vfi_warp_m2m(src.p + soff, src.c, swap, flow.p + foff, flow.c, dst.p + doff, dst.c, src.n, src.h, src.w, c, st);
}

int cube(vfi_m2m* m, hipStream_t st) {
    const int h = m->ench[3][0], w = m->ench[3][1];
    if (vfi_pool_mean(m->s3.p, 256, m->ph.p, 256, 2, h, w, 256, 1, st) || vfi_pool_mean(m->s3.p, 256, m->pw.p, 256, 2, h, w, 256, 2, st))
        return -1;
    // global mean = mean over rows of the row means (every row has w pixels)
    if (vfi_pool_mean(m->ph.p, 256, m->pc.p, 256, 2, h, 1, 256, 0, st)) return -1;
    if (run(m->cube[0], m->pc, 0, m->cc, 0, 4, 0.f, st) || run(m->cube[1], m->ph, 0, m->ch, 0, 4, 0.f, st) ||
        run(m->cube[2], m->pw, 0, m->cw, 0, 4, 0.f, st))
        return -1;
    return vfi_m2m_cube_apply(m->s3.p, 256, m->cc.p, m->ch.p, 16, m->cw.p, 16, m->enc[3].p, 768, 2, h, w, 256, st);
}

// EncDec's image pyramid (:722-735): c[l] = pyr[l](c[l-1]) written into its channel window of enc[l]; depends on the frames only
int image_pyramid(vfi_m2m* m, hipStream_t st) {
    const int coff[4] = {32, 64, 128, 256};   // channel offset of the image-pyramid feature c[l] inside enc[l]
    const Ten* src = &m->d0;
    int soff = 0;
    for (int l = 0; l < 4; ++l) {
        Ten* t;
        if (tmp(m, "pa", m->ench[l][0], m->ench[l][1], 16 << l, &t)) return -1;
        if (run(m->pyr[l][0], *src, soff, *t, 0, 3, 0.f, st) || run(m->pyr[l][1], *t, 0, m->enc[l], coff[l], 3, 0.f, st)) return -1;
        src = &m->enc[l], soff = coff[l];
    }
    return 0;
}

}  // namespace

extern "C" {

vfi_m2m_t* vfi_m2m_create(const float* const* tensors, const int64_t* numels, int n_tensors) {
    if (!tensors || !numels || n_tensors != 188) {
        set_error("vfi_m2m_create: expected the 188 state_dict tensors of M2M_PWC in m2m_spec.m2m_shapes() order, got %d", n_tensors);
        return nullptr;
    }
    vfi_m2m* m = new vfi_m2m();
    int k = 0;
    bool ok = true;
    auto take = [&](int64_t want) -> const float* {
        if (!ok) return nullptr;
        if (k >= n_tensors || numels[k] != want) {
            set_error("vfi_m2m_create: tensor %d has %lld elements, expected %lld", k, (long long)(k < n_tensors ? numels[k] : -1), (long long)want);
            ok = false;
            return nullptr;
        }
        return tensors[k++];
    };
    auto make = [&](Layer& L, int kind, const float* w, const float* b, int cout, int cin, int kk, int stride, int pad_mode,
                    const std::vector<int>* cmap, int cin_phys, const float* prelu) {
        if (!ok) return;
        L.kind = kind, L.stride = stride;
        L.h = vfi_conv_create_ex(kind, w, b, cout, cin, kk, stride, pad_mode, cmap ? cmap->data() : nullptr, cin_phys > 0 ? cin_phys : (cin + 7) / 8 * 8,
                                 prelu);
        if (!L.h) ok = false;
        else m->all.push_back(L.h);
    };
    const float* a = take(1);
    if (a) m->alpha = a[0];
    // PWC extractor: 3 x (sconv(2)-prelu, conv(3,replpad)-prelu, conv(3,replpad)-prelu), M2M_arch.py:415-446
    const int ext_cin[3] = {3, 32, 32};
    for (int s = 0; s < 3 && ok; ++s) {
        const float* w = take((int64_t)32 * ext_cin[s] * 4);
        const float* b = take(32);
        const float* sl = take(1);
        make(m->ext[s][0], 0, w, b, 32, ext_cin[s], 2, 2, 0, nullptr, 0, nullptr);
        if (sl) m->ext[s][0].slope = sl[0];
        for (int i = 1; i <= 2 && ok; ++i) {
            w = take((int64_t)32 * 32 * 9);
            b = take(32);
            sl = take(1);
            make(m->ext[s][i], 0, w, b, 32, 32, 3, 1, 1, nullptr, 0, nullptr);
            if (sl) m->ext[s][i].slope = sl[0];
        }
    }
    // PWC decoders, checkpoint order netFiv, netFou, netThr, netTwo, netOne = levels 4..0 (:449-503)
    for (int p = 0; p < 5 && ok; ++p) {
        const int lvl = 4 - p, cin0 = lvl == 4 ? 113 : 115;
        (void)take(1);   // netCostacti: PReLU on the cost volume, which is >= 0 — the identity
        const int chans[7] = {cin0, 128, 128, 96, 64, 32, 2};
        for (int i = 0; i < 6 && ok; ++i) {
            const float* w = take((int64_t)chans[i + 1] * chans[i] * 9);
            const float* b = take(chans[i + 1]);
            const float* sl = i < 5 ? take(1) : nullptr;
            make(m->dec[lvl][i], 0, w, b, chans[i + 1], chans[i], 3, 1, 1, nullptr, i == 0 ? DEC_CS : 0, nullptr);
            m->dec[lvl][i].slope = sl ? sl[0] : 0.f;
        }
    }
    // conv() helper: Conv2d(3, stride, 1) + PReLU(cout), :589-602
    auto cp = [&](Layer& L, int cin, int cout, int stride, const std::vector<int>* cmap, int cin_phys) {
        const float* w = take((int64_t)cout * cin * 9);
        const float* b = take(cout);
        const float* pr = take(cout);
        make(L, 0, w, b, cout, cin, 3, stride, 0, cmap, cin_phys, pr);
    };
    const int pyr_c[5] = {3, CC, 2 * CC, 4 * CC, 8 * CC};
    const std::vector<int> img_map = {2, 3, 4};   // the image channels of d0 = [flow 2 | image 3 | ...]
    for (int i = 0; i < 4 && ok; ++i) {
        cp(m->pyr[i][0], pyr_c[i], pyr_c[i + 1], 2, i == 0 ? &img_map : nullptr, i == 0 ? 8 : 0);
        cp(m->pyr[i][1], pyr_c[i + 1], pyr_c[i + 1], 1, nullptr, 0);
    }
    const int down_ci[4] = {8, 6 * CC, 12 * CC, 24 * CC}, down_co[4] = {2 * CC, 4 * CC, 8 * CC, 16 * CC};
    for (int i = 0; i < 4 && ok; ++i) {
        cp(m->down[i][0], down_ci[i], down_co[i], 2, nullptr, 0);
        cp(m->down[i][1], down_co[i], down_co[i], 1, nullptr, 0);
    }
    const int up_ci[4] = {48 * CC, 16 * CC, 8 * CC, 4 * CC}, up_co[4] = {8 * CC, 4 * CC, 2 * CC, CC};
    for (int i = 0; i < 4 && ok; ++i) {
        const float* w = take((int64_t)up_ci[i] * up_co[i] * 16);
        const float* b = take(up_co[i]);
        const float* pr = take(up_co[i]);
        make(m->up[i], 1, w, b, up_co[i], up_ci[i], 4, 2, 0, nullptr, 0, pr);
    }
    // conv (8 flow residuals) and conv_m (mask logit) read the same tensor: one layer with 9 outputs (:838-846)
    {
        const float* w8 = take((int64_t)8 * CC * 9);
        const float* b8 = take(8);
        const float* w1 = take((int64_t)1 * CC * 9);
        const float* b1 = take(1);
        if (ok) {
            std::vector<float> w(9 * CC * 9), b(9);
            memcpy(w.data(), w8, sizeof(float) * 8 * CC * 9);
            memcpy(w.data() + 8 * CC * 9, w1, sizeof(float) * CC * 9);
            memcpy(b.data(), b8, sizeof(float) * 8);
            b[8] = b1[0];
            make(m->head, 0, w.data(), b.data(), 9, CC, 3, 1, 0, nullptr, 0, nullptr);
        }
    }
    const int cube_co[3] = {16 * 16 * CC, 16, 16};
    for (int i = 0; i < 3 && ok; ++i) {
        const float* w = take((int64_t)cube_co[i] * 16 * CC);
        const float* b = take(cube_co[i]);
        make(m->cube[i], 0, w, b, cube_co[i], 16 * CC, 1, 1, 0, nullptr, 0, nullptr);
    }
    if (!ok || k != n_tensors) {
        if (ok) set_error("vfi_m2m_create: consumed %d of %d tensors", k, n_tensors);
        vfi_m2m_destroy(m);
        return nullptr;
    }
    return m;
}

void vfi_m2m_destroy(vfi_m2m_t* m) {
    if (!m) return;
    if (m->side) {
        (void)hipStreamSynchronize(m->side);
        (void)hipEventDestroy(m->ev_fork);
        (void)hipEventDestroy(m->ev_join);
        (void)hipStreamDestroy(m->side);
    }
    for (vfi_conv_t* c : m->all) vfi_conv_destroy(c);
    free_workspace(m);
    delete m;
}

int64_t vfi_m2m_workspace_bytes(vfi_m2m_t* m) { return m ? m->owned_bytes : 0; }

int vfi_m2m_release_workspace(vfi_m2m_t* m) {
    VFI_REQUIRE(m, "vfi_m2m_release_workspace: null handle");
    VFI_CHECK_HIP(hipDeviceSynchronize());
    free_workspace(m);
    return 0;
}

int vfi_m2m_prepare(vfi_m2m_t* m, const float* frame0_dev, const float* frame1_dev, int C, int H, int W, void* stream) {
    VFI_REQUIRE(m && frame0_dev && frame1_dev && C >= 3 && H > 0 && W > 0, "vfi_m2m_prepare: bad arguments");
    if (int rc = ensure_workspace(m, H, W)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int Hp = m->Hp, Wp = m->Wp;
    m->prepared = false;
    if (vfi_m2m_normalize(frame0_dev, frame1_dev, C, H, W, Hp, Wp, m->d0.p, 8, 2, m->stats, m->ws, 16384, st)) return -1;
    if (vfi_m2m_image4(m->d0.p, 8, m->img4.p, Hp, Wp, st)) return -1;
    // fork: the c features (image pyramid through pyr[l][0..1], EncDec.forward :722-735) read channels 2..4 of d0 = the normalised frames,
    // which exist now.  (Channels 0..1 and 5..7 of d0 are written later on `st` — the flow, the warped partner — while the side stream may
    // still read the 8-channel pixels: those channels meet zero weights in pyr[0][0], on either stream.)
    const bool forked = option(kOptM2mSide) != 0;
    if (forked) {
        if (!m->side) {
            VFI_CHECK_HIP(hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking));
            VFI_CHECK_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
            VFI_CHECK_HIP(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
        }
        VFI_CHECK_HIP(hipEventRecord(m->ev_fork, st));
        VFI_CHECK_HIP(hipStreamWaitEvent(m->side, m->ev_fork, 0));
        if (image_pyramid(m, m->side)) return -1;
        VFI_CHECK_HIP(hipEventRecord(m->ev_join, m->side));
    }
    // ---- flow network on half-resolution images (:936-939, bidir :521-546)
    if (resize(m->d0, 2, m->imh, 0, 3, 1.0f, st)) return -1;
    const Ten* src = &m->imh;
    for (int k = 0; k < 3; ++k) {
        const int hh = m->dech[k][0], ww = m->dech[k][1];
        Ten *ta, *tb;
        if (tmp(m, "ea", hh, ww, 32, &ta) || tmp(m, "eb", hh, ww, 32, &tb)) return -1;
        if (run(m->ext[k][0], *src, 0, *ta, 0, 1, m->ext[k][0].slope, st) || run(m->ext[k][1], *ta, 0, *tb, 0, 1, m->ext[k][1].slope, st) ||
            run(m->ext[k][2], *tb, 0, m->decb[k], 0, 1, m->ext[k][2].slope, st))
            return -1;
        src = &m->decb[k];
    }
    for (int k = 3; k <= 4; ++k)   // netFou / netFiv features: avg_pool2d(2, 2) (:438-444)
        if (vfi_avgpool2(m->decb[k - 1].p, DEC_CS, m->decb[k].p, DEC_CS, 2, m->dech[k - 1][0], m->dech[k - 1][1], 32, st)) return -1;
    for (int k = 4; k >= 0; --k) {
        const int hh = m->dech[k][0], ww = m->dech[k][1];
        Ten& d = m->decb[k];
        if (k == 4) {
            if (vfi_costvol9x9(d.p, DEC_CS, d.p, DEC_CS, 1, d.p, 2, hh, ww, 32, DEC_CS, 32, st)) return -1;
        } else {
            if (resize(m->flow[k + 1], 0, d, FLOW_OFF, 2, 2.0f, st)) return -1;
            Ten* wb;
            if (tmp(m, "wb", hh, ww, 32, &wb)) return -1;
            if (warp(d, 0, 32, d, FLOW_OFF, *wb, 0, st)) return -1;
            if (vfi_costvol9x9(d.p, DEC_CS, wb->p, 32, 0, d.p, 2, hh, ww, 32, DEC_CS, 32, st)) return -1;
        }
        Ten *ta, *tb;
        if (tmp(m, "da", hh, ww, 128, &ta) || tmp(m, "db", hh, ww, 128, &tb)) return -1;
        const Layer* cv = m->dec[k];
        if (run(cv[0], d, 0, *ta, 0, 1, cv[0].slope, st) || run(cv[1], *ta, 0, *tb, 0, 1, cv[1].slope, st) ||
            run(cv[2], *tb, 0, *ta, 0, 1, cv[2].slope, st) || run(cv[3], *ta, 0, *tb, 0, 1, cv[3].slope, st) ||
            run(cv[4], *tb, 0, *ta, 0, 1, cv[4].slope, st))
            return -1;
        if (k == 4) {
            if (run(cv[5], *ta, 0, m->flow[k], 0, 0, 0.f, st)) return -1;
        } else {
            if (run(cv[5], *ta, 0, m->flow[k], 0, 0, 0.f, st, &d, FLOW_OFF)) return -1;   // flow + netMain(...), :503
        }
    }
    // ---- motion refinement (MotionRefineNet.forward :866-890, EncDec.forward :718-848)
    if (resize(m->flow[0], 0, m->d0, 0, 2, (float)RATIO, st)) return -1;
    if (forked) VFI_CHECK_HIP(hipStreamWaitEvent(st, m->ev_join, 0));
    else if (image_pyramid(m, st)) return -1;
    if (vfi_m2m_warp_image4(m->img4.p, m->d0.p, 8, m->d0.p + 5, 8, Hp, Wp, st)) return -1;
    src = &m->d0;
    const Ten* flow_src = &m->d0;
    for (int l = 0; l < 4; ++l) {
        Ten* t;
        if (tmp(m, "dn", m->ench[l][0], m->ench[l][1], 32 << l, &t)) return -1;
        if (run(m->down[l][0], *src, 0, *t, 0, 3, 0.f, st)) return -1;
        const Ten& dst = l < 3 ? m->enc[l] : m->s3;
        if (run(m->down[l][1], *t, 0, dst, 0, 3, 0.f, st)) return -1;
        if (resize(*flow_src, 0, m->fl[l + 1], 0, 2, 0.5f, st)) return -1;
        flow_src = &m->fl[l + 1];
        if (l == 3 && cube(m, st)) return -1;
        const int nfeat = 48 << l;   // s + c channels of this level
        if (warp(m->enc[l], 0, nfeat, m->fl[l + 1], 0, m->enc[l], nfeat, st)) return -1;
        src = &m->enc[l];
    }
    // up path: x is written over the (already consumed) c / warp slots of the level above -> cat(s, x) is a window
    if (run(m->up[0], m->enc[3], 0, m->enc[2], 128, 3, 0.f, st) || run(m->up[1], m->enc[2], 0, m->enc[1], 64, 3, 0.f, st) ||
        run(m->up[2], m->enc[1], 0, m->enc[0], 32, 3, 0.f, st) || run(m->up[3], m->enc[0], 0, m->xf, 0, 3, 0.f, st))
        return -1;
    if (run(m->head, m->xf, 0, m->r, 0, 0, 0.f, st)) return -1;
    if (vfi_m2m_photo_tiles(m->d0.p, 8, m->r.p, 12, m->alpha, m->img4.p, m->tf.p, m->e.p, m->tile_ranges, m->smax, Hp, Wp, st)) return -1;
    m->prepared = true;
    return 0;
}

int vfi_m2m_render(vfi_m2m_t* m, float t, float* out_dev, void* stream) {
    VFI_REQUIRE(m && out_dev, "vfi_m2m_render: bad arguments");
    VFI_REQUIRE(m->prepared, "vfi_m2m_render: no prepared frame pair (call vfi_m2m_prepare first)");
    hipStream_t st = (hipStream_t)stream;
    const int Hp = m->Hp, Wp = m->Wp;
    // one kernel: splat inputs, the 8 summation splats and forwarp_mframe_mask's combine per 32x32 tile of the frame (m2m_render.hip)
    if (option(kOptM2mFused) && t >= 0.f && t <= 1.f)
        return vfi_m2m_render_fused(m->img4.p, m->tf.p, m->e.p, m->tile_ranges, m->smax, m->stats, t, out_dev, Hp, Wp, m->H, m->W, st);
    // the three-step form (A/B option m2m_fused = 0, and timesteps outside [0, 1]); its buffers are allocated at first use
    if (!m->sin.p && (alloc_ten(m, m->sin, 8, Hp, Wp, 4, st) || alloc_ten(m, m->sfl, 8, Hp, Wp, 2, st) || alloc_ten(m, m->sout, 8, Hp, Wp, 4, st))) return -1;
    if (vfi_m2m_splat_inputs(m->d0.p, 8, m->tf.p, m->e.p, t, m->sin.p, m->sfl.p, Hp, Wp, st)) return -1;
    if (vfi_softsplat_sum(m->sin.p, m->sfl.p, m->sout.p, 8, Hp, Wp, 4, st)) return -1;
    return vfi_m2m_combine(m->sout.p, m->d0.p, 8, m->stats, t, out_dev, Hp, Wp, m->H, m->W, st);
}

#ifdef VFI_TEST_TAPS
// test tap (include/vfi_hip_test.h, libvfi_hip_test.so only): internal tensors of the LAST prepare — what 0: PWC flows at 1/4 of the padded size
// [2,Hp/4,Wp/4,2]; 1: d0 [2,Hp,Wp,8] (refined-flow base | normalised image | warped partner); 2: r [2,Hp,Wp,12] (8 residuals | mask)
int64_t vfi_m2m_debug_read(vfi_m2m_t* m, int what, float* host_buf, int64_t cap) {
    if (!m || !m->prepared || what < 0 || what > 2) {
        set_error("vfi_m2m_debug_read: nothing prepared / bad selector %d", what);
        return -1;
    }
    const Ten& t = what == 0 ? m->flow[0] : (what == 1 ? m->d0 : m->r);
    const int64_t n = (int64_t)t.n * t.h * t.w * t.c;
    if (n > cap || hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_buf, t.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("vfi_m2m_debug_read: buffer too small (%lld needed) or copy failed", (long long)n);
        return -1;
    }
    return n;
}
#endif  // VFI_TEST_TAPS

}  // extern "C"
