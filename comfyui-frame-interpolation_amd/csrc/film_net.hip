// FILM interpolator as a C-side object ("same triplet for FILM"): weights resident, workspace owned, the whole
// launch sequence of one frame pair issued by ONE C call — vfi_film_create / vfi_film_forward / vfi_film_destroy.
//
// Replaces Interpolator.debug_forward (vfi_models/film/film_arch.py:401-455, time = 0.5) with its sub-modules:
// build_image_pyramid :655-674, FeatureExtractor / SubTreeExtractor :83-162, PyramidFlowEstimator / FlowEstimator
// :500-617, flow_pyramid_synthesis :745-755, pyramid_warp :758-772, concatenate_pyramids + multiply_pyramid :727-742,
// Fusion :219-292.  Built on the library's own generic entry points (vfi_conv_forward on the fp32 matrix cores,
// vfi_avgpool2, vfi_warp_film, vfi_resize_bilinear, vfi_upsample_nearest, vfi_axpby): every torch.cat along channels is
// a channel-window offset, physical channel positions are translated once through chan_map at weight-pack time.
// (Round 1 issued the same sequence from Python, ~330 ctypes calls per pair; the Python FilmEngine now wraps this object.)
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

constexpr int PYR = 7, FUS = 5, SUB = 4, FILTERS = 64;
const int kFlowFilters[4] = {32, 64, 128, 256};

int r8(int n) { return (n + 7) / 8 * 8; }
int feat_channels(int level) {   // 64, 192, 448, 960, 960, ...
    int s = 0;
    for (int j = 0; j <= std::min(level, SUB - 1); ++j) s += FILTERS << j;
    return s;
}

struct Ten {   // [h][w][c] fp32, zero-initialised at allocation (padded channel positions must hold finite values)
    float* p = nullptr;
    int h = 0, w = 0, c = 0;
};

struct Layer {
    vfi_conv_t* h = nullptr;
    int cout = 0, cin_phys = 0;
};

// reference channel order of one aligned-pyramid level [imgA 3, featA F, imgB 3, featB F, bflow 2, fflow 2]
// -> physical order [imgA 3, 0, featA F | imgB 3, 0, featB F | bflow 2, fflow 2 | zero pad to x8]
std::vector<int> aligned_map(int F) {
    std::vector<int> m;
    for (int half = 0; half < 2; ++half) {
        const int base = half * (4 + F);
        for (int c = 0; c < 3; ++c) m.push_back(base + c);
        for (int c = 0; c < F; ++c) m.push_back(base + 4 + c);
    }
    for (int c = 0; c < 4; ++c) m.push_back(2 * (4 + F) + c);
    return m;
}

}  // namespace

struct vfi_film {
    Layer ext[SUB][2];
    Layer pred[4][5];          // [min(level, 3)][conv]
    Layer fuse[4][3];
    Layer fuse_up2[4];         // fuse[f][0] in its "up-sample x2 first" form (vfi_conv_create_up2x2): used where the level above is exactly half the size
    int fuse_nf[4] = {0, 0, 0, 0};
    Layer out_conv;
    int cal[FUS] = {0, 0, 0, 0, 0};
    // workspace for the current frame size
    int H = 0, W = 0;
    int hw[PYR][2];
    Ten img[2][PYR], tw[2][PYR], flow[2][PYR], vres[2][PYR], vup[2][PYR], al[FUS];      // vres / vup: per flow direction (r6: the two run side by side)
    // r6: image 1's feature extraction and the backward flow pyramid run on this side stream (another hardware queue) beside image 0's and
    // the forward one: two independent halves of the network up to the fusion, each with coarse levels that fill a fraction of the device
    bool two_streams = true;       // vfi_film_two_streams
    hipStream_t side = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::map<std::tuple<std::string, int, int, int>, Ten> scratch;
    std::vector<float*> owned;
};

namespace {

int alloc_ten(vfi_film* n, Ten& t, int h, int w, int c, hipStream_t st = nullptr) {
    t.h = h, t.w = w, t.c = c;
    const size_t bytes = (size_t)h * w * c * sizeof(float);
    VFI_CHECK_HIP(hipMalloc((void**)&t.p, bytes));
    n->owned.push_back(t.p);
    // zero fill ordered with the forward's kernels: a NULL-stream memset is not ordered against a non-blocking side stream (torch's)
    // and could clear a lazily allocated scratch tensor AFTER its first producer ran
    VFI_CHECK_HIP(hipMemsetAsync(t.p, 0, bytes, st));
    if (!st) VFI_CHECK_HIP(hipStreamSynchronize(nullptr));
    return 0;
}

void free_workspace(vfi_film* n) {
    for (float* p : n->owned) (void)hipFree(p);
    n->owned.clear();
    n->scratch.clear();
    n->H = n->W = 0;
}

int ensure_workspace(vfi_film* n, int H, int W) {
    if (n->H == H && n->W == W) return 0;
    VFI_REQUIRE(H >= 64 && W >= 64, "vfi_film_forward: FILM needs 7 pyramid levels (>= 64 px per side), got %dx%d", H, W);
    VFI_CHECK_HIP(hipDeviceSynchronize());
    free_workspace(n);
    n->hw[0][0] = H, n->hw[0][1] = W;
    for (int l = 1; l < PYR; ++l) n->hw[l][0] = n->hw[l - 1][0] / 2, n->hw[l][1] = n->hw[l - 1][1] / 2;
    for (int l = 0; l < PYR; ++l) {
        const int h = n->hw[l][0], w = n->hw[l][1], F = feat_channels(l);
        for (int k = 0; k < 2; ++k) {
            if (alloc_ten(n, n->img[k][l], h, w, 8) || alloc_ten(n, n->tw[k][l], h, w, 4 + 2 * F) || alloc_ten(n, n->flow[k][l], h, w, 2)) return -1;
        }
        for (int d = 0; d < 2; ++d)
            if (alloc_ten(n, n->vres[d][l], h, w, 2) || alloc_ten(n, n->vup[d][l], h, w, 2)) return -1;
        if (l < FUS && alloc_ten(n, n->al[l], h, w, n->cal[l] + (l < 4 ? FILTERS << std::min(l, 3) : 0))) return -1;
    }
    n->H = H, n->W = W;
    return 0;
}

int tmp(vfi_film* n, const char* name, int h, int w, int c, Ten** out) {
    auto key = std::make_tuple(std::string(name), h, w, c);
    auto it = n->scratch.find(key);
    if (it == n->scratch.end()) {
        Ten t;
        if (alloc_ten(n, t, h, w, c)) return -1;
        it = n->scratch.emplace(key, t).first;
    }
    *out = &it->second;
    return 0;
}

int conv(const Layer& L, const Ten& src, int src_off, const Ten& dst, int dst_off, int h, int w, int act, hipStream_t st) {
    return vfi_conv_forward(L.h, src.p + src_off, src.c, dst.p + dst_off, dst.c, 1, h, w, act, 0.2f, st);
}
int conv_raw(const Layer& L, const Ten& src, int src_off, float* dst, int dst_cs, int h, int w, int act, hipStream_t st) {
    return vfi_conv_forward(L.h, src.p + src_off, src.c, dst, dst_cs, 1, h, w, act, 0.2f, st);
}
int ax(const float* a, int a_cs, const float* b, int b_cs, float* out, int out_cs, int h, int w, int c, float alpha, float beta,
       hipStream_t st) {
    return vfi_axpby(a, a_cs, b, b_cs, out, out_cs, (int64_t)h * w, c, alpha, beta, st);
}

// FeatureExtractor on image pyramid k: the cascaded feature pyramid goes into tw[k][*][..., 4:]
int extract(vfi_film* n, int k, hipStream_t st) {
    for (int i = 0; i < PYR; ++i) {
        const int depth = std::min(PYR - i, SUB);
        const Ten* src = &n->img[k][i];
        int src_off = 0;
        for (int j = 0; j < depth; ++j) {
            const int lvl = i + j, h = n->hw[lvl][0], w = n->hw[lvl][1], c = FILTERS << j;
            Ten* mid;
            if (tmp(n, k ? "ext_mid1" : "ext_mid0", h, w, c, &mid)) return -1;
            if (conv(n->ext[j][0], *src, src_off, *mid, 0, h, w, 1, st)) return -1;
            int slot_off = 4;
            for (int q = 0; q < j; ++q) slot_off += FILTERS << q;    // position of sub[.][j] inside level lvl's features
            if (conv(n->ext[j][1], *mid, 0, n->tw[k][lvl], slot_off, h, w, 1, st)) return -1;
            if (j < depth - 1) {
                const int h2 = n->hw[lvl + 1][0], w2 = n->hw[lvl + 1][1];
                Ten* pooled;
                if (tmp(n, k ? "ext_pool1" : "ext_pool0", h2, w2, c, &pooled)) return -1;
                if (vfi_avgpool2(n->tw[k][lvl].p + slot_off, n->tw[k][lvl].c, pooled->p, c, 1, h, w, c, st)) return -1;
                src = pooled, src_off = 0;
            }
        }
    }
    return 0;
}

// PyramidFlowEstimator(feature pyramid a, feature pyramid b) -> flow[d][l] = synthesised flow per level
int predict(vfi_film* n, int a, int b, int d, hipStream_t st) {
    for (int l = PYR - 1; l >= 0; --l) {
        const int h = n->hw[l][0], w = n->hw[l][1], F = feat_channels(l);
        // torch.cat([features_a, warp(features_b)]) (film_arch.py:606-611) is a channel WINDOW: pyramid a's tensor carries a slot of F
        // channels behind its own features that the partner's warped features are written into — the flow estimator's first
        // convolution reads [4, 4 + 2F) of it.  (r2 copied features_a next to the warp result: 14 copies per pair, ~1.5 ms at 1080p.)
        Ten& pair = n->tw[a][l];
        const int PO = 4;          // channel offset of the window
        if (l == PYR - 1) {
            if (ax(n->tw[b][l].p + 4, n->tw[b][l].c, nullptr, 0, pair.p + PO + F, pair.c, h, w, F, 1.f, 0.f, st)) return -1;
        } else {
            const int h1 = n->hw[l + 1][0], w1 = n->hw[l + 1][1];
            if (vfi_resize_bilinear(n->flow[d][l + 1].p, 2, n->vup[d][l].p, 2, 1, h1, w1, h, w, 2, 2.0f, st)) return -1;
            if (vfi_warp_film(n->tw[b][l].p + 4, n->tw[b][l].c, n->vup[d][l].p, 2, 1.0f, pair.p + PO + F, pair.c, 1, h, w, F, st)) return -1;
        }
        const Layer* convs = n->pred[std::min(l, 3)];
        const int nf = kFlowFilters[std::min(l, 3)];
        Ten *t0, *t1, *t2;
        if (tmp(n, d ? "fe0b" : "fe0", h, w, nf, &t0) || tmp(n, d ? "fe1b" : "fe1", h, w, nf, &t1) || tmp(n, d ? "fe2b" : "fe2", h, w, r8(nf / 2), &t2)) return -1;
        if (conv(convs[0], pair, PO, *t0, 0, h, w, 1, st) || conv(convs[1], *t0, 0, *t1, 0, h, w, 1, st) ||
            conv(convs[2], *t1, 0, *t0, 0, h, w, 1, st) || conv(convs[3], *t0, 0, *t2, 0, h, w, 1, st))
            return -1;
        if (l == PYR - 1) {
            if (conv(convs[4], *t2, 0, n->flow[d][l], 0, h, w, 0, st)) return -1;
        } else {
            if (conv(convs[4], *t2, 0, n->vres[d][l], 0, h, w, 0, st)) return -1;
            if (ax(n->vres[d][l].p, 2, n->vup[d][l].p, 2, n->flow[d][l].p, 2, h, w, 2, 1.f, 1.f, st)) return -1;   // v = v_residual + v
        }
    }
    return 0;
}

}  // namespace

extern "C" {

vfi_film_t* vfi_film_create(const float* const* tensors, const int64_t* numels, int n_tensors) {
    if (!tensors || !numels || n_tensors != 82) {
        set_error("vfi_film_create: expected the 82 state_dict tensors of the FILM Interpolator in film_spec.film_shapes() order, got %d", n_tensors);
        return nullptr;
    }
    vfi_film* net = new vfi_film();
    int k = 0;
    bool ok = true;
    auto make = [&](Layer& L, int cout, int cin, int kk, const std::vector<int>* cmap, int cin_phys) {
        if (!ok) return;
        const int64_t wn = (int64_t)cout * cin * kk * kk;
        if (k + 1 >= n_tensors || numels[k] != wn || numels[k + 1] != cout) {
            set_error("vfi_film_create: tensor %d has %lld elements, expected %lld (weight of a %d->%d %dx%d conv)", k,
                      (long long)numels[k], (long long)wn, cin, cout, kk, kk);
            ok = false;
            return;
        }
        L.cout = cout;
        L.cin_phys = cin_phys > 0 ? cin_phys : r8(cin);
        L.h = vfi_conv_create(tensors[k], tensors[k + 1], cout, cin, kk, kk, cmap ? cmap->data() : nullptr, L.cin_phys);
        k += 2;
        if (!L.h) ok = false;
    };
    // extract.extract_sublevels.convs.{i}.{0,1}.0
    int cin = 3;
    for (int i = 0; i < SUB; ++i) {
        const int c = FILTERS << i;
        make(net->ext[i][0], c, cin, 3, nullptr, 0);
        make(net->ext[i][1], c, c, 3, nullptr, 0);
        cin = c;
    }
    // predict_flow._predictor (levels >= 3), then _predictors.0/1/2 = levels 2/1/0 (film_arch.py:562-563)
    const int pred_level[4] = {3, 2, 1, 0};
    for (int e = 0; e < 4; ++e) {
        const int lvl = pred_level[e], nf = kFlowFilters[lvl];
        int c = 2 * feat_channels(lvl);
        for (int i = 0; i < 3; ++i) {
            make(net->pred[lvl][i], nf, c, 3, nullptr, 0);
            c = nf;
        }
        make(net->pred[lvl][3], nf / 2, nf, 1, nullptr, 0);
        make(net->pred[lvl][4], 2, nf / 2, 1, nullptr, 0);
    }
    for (int l = 0; l < FUS; ++l) net->cal[l] = r8(2 * (4 + feat_channels(l)) + 4);
    make(net->out_conv, 3, FILTERS, 1, nullptr, 0);
    for (int f = 0; f < 4; ++f) {
        const int lvl = 3 - f, nf = FILTERS << std::min(lvl, 3);
        const int below = f == 0 ? 2 * (3 + feat_channels(lvl + 1)) + 4 : FILTERS << std::min(lvl + 1, 3);
        const int skip = 2 * (3 + feat_channels(lvl)) + 4;
        net->fuse_nf[f] = nf;
        if (f == 0) {
            const std::vector<int> m = aligned_map(feat_channels(4));
            make(net->fuse[f][0], nf, below, 2, &m, net->cal[4]);
            if (ok) {
                net->fuse_up2[f].cout = nf, net->fuse_up2[f].cin_phys = net->cal[4];
                net->fuse_up2[f].h = vfi_conv_create_up2x2(tensors[k - 2], tensors[k - 1], nf, below, m.data(), net->cal[4]);
                if (!net->fuse_up2[f].h) ok = false;
            }
        } else {
            make(net->fuse[f][0], nf, below, 2, nullptr, 0);
            if (ok) {
                net->fuse_up2[f].cout = nf, net->fuse_up2[f].cin_phys = r8(below);
                net->fuse_up2[f].h = vfi_conv_create_up2x2(tensors[k - 2], tensors[k - 1], nf, below, nullptr, r8(below));
                if (!net->fuse_up2[f].h) ok = false;
            }
        }
        std::vector<int> m = aligned_map(feat_channels(lvl));
        for (int c = 0; c < nf; ++c) m.push_back(net->cal[lvl] + c);
        (void)skip;
        make(net->fuse[f][1], nf, skip + nf, 3, &m, net->cal[lvl] + nf);
        make(net->fuse[f][2], nf, nf, 3, nullptr, 0);
    }
    if (!ok || k != n_tensors) {
        if (ok) set_error("vfi_film_create: consumed %d of %d tensors", k, n_tensors);
        vfi_film_destroy(net);
        return nullptr;
    }
    return net;
}

void vfi_film_destroy(vfi_film_t* net) {
    if (!net) return;
    for (auto& st : net->ext)
        for (Layer& L : st) vfi_conv_destroy(L.h);
    for (auto& p : net->pred)
        for (Layer& L : p) vfi_conv_destroy(L.h);
    for (auto& f : net->fuse)
        for (Layer& L : f) vfi_conv_destroy(L.h);
    for (Layer& L : net->fuse_up2) vfi_conv_destroy(L.h);
    vfi_conv_destroy(net->out_conv.h);
    if (net->side) {
        stream_give_back(net->side);      // (drained there; kept for the next object: a node that rebuilds its model per call creates no streams per call)
        for (hipEvent_t e : net->ev) (void)hipEventDestroy(e);
    }
    free_workspace(net);
    delete net;
}

int vfi_film_two_streams(vfi_film_t* net, int on) {
    VFI_REQUIRE(net, "vfi_film_two_streams: null handle");
    const int before = net->two_streams ? 1 : 0;
    net->two_streams = on != 0;
    return before;
}

int vfi_film_release_workspace(vfi_film_t* net) {
    VFI_REQUIRE(net, "vfi_film_release_workspace: null handle");
    VFI_CHECK_HIP(hipDeviceSynchronize());
    free_workspace(net);
    return 0;
}

int vfi_film_forward(vfi_film_t* net, const float* x0_dev, const float* x1_dev, int C, int H, int W, float* out_dev, int clamp,
                     void* stream) {
    VFI_REQUIRE(net && x0_dev && x1_dev && out_dev && C >= 3, "vfi_film_forward: bad arguments");
    if (int rc = ensure_workspace(net, H, W)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const float* xs[2] = {x0_dev, x1_dev};
    // Two streams (option film_side): half k of the network — image k's pyramid and features, then the flow pyramid of direction k — runs
    // on hs[k]; the halves meet twice: each flow estimator reads the OTHER image's features, and the fusion reads everything.
    const bool two = net->two_streams && option(kOptFilmSide) != 0;
    if (two && !net->side) {
        net->side = stream_apart_from(st);
        VFI_REQUIRE(net->side, "vfi_film_forward: no side stream");
        for (hipEvent_t& e : net->ev) VFI_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipStream_t hs[2] = {st, two ? net->side : st};
    if (two) {
        VFI_CHECK_HIP(hipEventRecord(net->ev[0], st));                 // the caller's frames (and the previous call's readers) are ordered on st
        VFI_CHECK_HIP(hipStreamWaitEvent(net->side, net->ev[0], 0));
    }
    for (int k = 0; k < 2; ++k) {
        hipStream_t s = hs[k];
        if (ax(xs[k], C, nullptr, 0, net->img[k][0].p, 8, H, W, 3, 1.f, 0.f, s)) return -1;
        for (int l = 1; l < PYR; ++l)   // build_image_pyramid
            if (vfi_avgpool2(net->img[k][l - 1].p, 8, net->img[k][l].p, 8, 1, net->hw[l - 1][0], net->hw[l - 1][1], 4, s)) return -1;
        for (int l = 0; l < PYR; ++l)   // image part of the to-warp pyramids
            if (ax(net->img[k][l].p, 8, nullptr, 0, net->tw[k][l].p, net->tw[k][l].c, net->hw[l][0], net->hw[l][1], 4, 1.f, 0.f, s)) return -1;
        if (extract(net, k, s)) return -1;
    }
    if (two) {      // each direction's estimator warps the partner's features into its own pyramid's slots
        VFI_CHECK_HIP(hipEventRecord(net->ev[1], st));
        VFI_CHECK_HIP(hipEventRecord(net->ev[2], net->side));
        VFI_CHECK_HIP(hipStreamWaitEvent(st, net->ev[2], 0));
        VFI_CHECK_HIP(hipStreamWaitEvent(net->side, net->ev[1], 0));
    }
    if (predict(net, 0, 1, 0, hs[0]) || predict(net, 1, 0, 1, hs[1])) return -1;   // forward / backward residual flow pyramids
    if (two) {
        VFI_CHECK_HIP(hipEventRecord(net->ev[3], net->side));
        VFI_CHECK_HIP(hipStreamWaitEvent(st, net->ev[3], 0));
    }
    // aligned pyramid: [warp(pyr0, 0.5*bwd) | warp(pyr1, 0.5*fwd) | 0.5*bwd | 0.5*fwd]
    for (int l = 0; l < FUS; ++l) {
        const int h = net->hw[l][0], w = net->hw[l][1], F = feat_channels(l);
        Ten& al = net->al[l];
        const Ten* srcs[2] = {&net->tw[0][l], &net->tw[1][l]};
        const Ten* fls[2] = {&net->flow[1][l], &net->flow[0][l]};
        for (int half = 0; half < 2; ++half)
            if (vfi_warp_film(srcs[half]->p, srcs[half]->c, fls[half]->p, 2, 0.5f, al.p + half * (4 + F), al.c, 1, h, w, 4 + F, st)) return -1;
        if (ax(net->flow[1][l].p, 2, nullptr, 0, al.p + 2 * (4 + F), al.c, h, w, 2, 0.5f, 0.f, st)) return -1;
        if (ax(net->flow[0][l].p, 2, nullptr, 0, al.p + 2 * (4 + F) + 2, al.c, h, w, 2, 0.5f, 0.f, st)) return -1;
    }
    // Fusion
    const Ten* cur = &net->al[4];
    int net_c = net->cal[4], nh = net->hw[4][0], nw = net->hw[4][1];
    for (int f = 0; f < 4; ++f) {
        const int lvl = 3 - f, h = net->hw[lvl][0], w = net->hw[lvl][1], nf = net->fuse_nf[f];
        Ten *up, *t, *o;
        if (h == 2 * nh && w == 2 * nw) {
            // an exact x2: nearest up-sampling + the 2x2 'same' convolution as one layer on the low-resolution tensor (9 tap blocks for 16)
            if (vfi_conv_forward(net->fuse_up2[f].h, cur->p, cur->c, net->al[lvl].p + net->cal[lvl], net->al[lvl].c, 1, nh, nw, 0, 0.2f, st)) return -1;
        } else {
            if (tmp(net, "fuse_up", h, w, net_c, &up)) return -1;
            if (vfi_upsample_nearest(cur->p, cur->c, up->p, net_c, 1, nh, nw, h, w, net_c, st)) return -1;
            if (conv(net->fuse[f][0], *up, 0, net->al[lvl], net->cal[lvl], h, w, 0, st)) return -1;
        }
        if (tmp(net, "fuse_t", h, w, nf, &t) || tmp(net, "fuse_o", h, w, nf, &o)) return -1;
        if (conv(net->fuse[f][1], net->al[lvl], 0, *t, 0, h, w, 1, st) || conv(net->fuse[f][2], *t, 0, *o, 0, h, w, 1, st)) return -1;
        cur = o, net_c = nf, nh = h, nw = w;
    }
    return conv_raw(net->out_conv, *cur, 0, out_dev, 3, H, W, clamp ? 2 : 0, st);
}

#ifdef VFI_TEST_TAPS
// test tap (include/vfi_hip_test.h, libvfi_hip_test.so only): the synthesised flow pyramid of the LAST forward, direction d (0 forward, 1 backward), level l
int64_t vfi_film_debug_read_flow(vfi_film_t* net, int d, int level, float* host_buf, int64_t cap) {
    if (!net || net->H == 0 || d < 0 || d > 1 || level < 0 || level >= PYR) {
        set_error("vfi_film_debug_read_flow: nothing to read (d=%d level=%d)", d, level);
        return -1;
    }
    const Ten& t = net->flow[d][level];
    const int64_t n = (int64_t)t.h * t.w * 2;
    if (n > cap || hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_buf, t.p, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("vfi_film_debug_read_flow: buffer too small (%lld needed) or copy failed", (long long)n);
        return -1;
    }
    return n;
}
#endif  // VFI_TEST_TAPS

}  // extern "C"
