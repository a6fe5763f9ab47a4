"""CPU: every per-element body of csrc/gmfss_bodies.h (the code the MI355X kernels of csrc/gmfss_ops.hip execute), run on the
host through tests/hostcheck and compared with the torch expression of the reference it replaces
(vfi_models/gmfss_fortuna/GMFSS_Fortuna_union_arch.py; restated in oracle/gmfss_oracle.py)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import hostcheck
from oracle import gmfss_oracle as G


@pytest.fixture(scope="module")
def lib():
    return hostcheck.load()


def P(t, off=0):
    return t.data_ptr() + 4 * off


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def ok(rc):
    assert rc == 0, rc


def test_pad_normalize_prelu_gelu_tanh_clamp(lib):
    torch.manual_seed(0)
    fr = torch.rand(37, 53, 4)
    out = torch.full((64, 64, 8), 9.0)
    ok(lib.vfi_pad_rgb(P(fr), 4, 37, 53, P(out), 8, 64, 64, None))
    want = F.pad(fr[..., :3].permute(2, 0, 1), (0, 11, 0, 27)).permute(1, 2, 0)
    assert torch.equal(out[..., :3], want) and (out[..., 3:] == 9.0).all()
    x = torch.randn(2, 9, 11, 8)
    y = torch.zeros(2, 9, 11, 8)
    mean, std = (C.c_float * 3)(0.485, 0.456, 0.406), (C.c_float * 3)(0.229, 0.224, 0.225)
    ok(lib.vfi_normalize_channels(P(x, 1), 8, P(y, 2), 8, 3, 2 * 9 * 11, mean, std, None))
    m, s = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    assert torch.equal(y[..., 2:5], (x[..., 1:4] - m) / s)
    ok(lib.vfi_prelu_scalar(P(x, 2), 8, P(y), 8, 5, 2 * 9 * 11, 0.25, None))
    assert torch.equal(y[..., :5], F.prelu(x[..., 2:7], torch.tensor([0.25])))
    g = x.clone()
    ok(lib.vfi_gelu(P(g, 3), 8, 4, 2 * 9 * 11, None))
    assert (g[..., 3:7] - F.gelu(x[..., 3:7])).abs().max() <= 1e-6 and torch.equal(g[..., :3], x[..., :3])
    t = x.clone()
    ok(lib.vfi_tanh_scale(P(t), 8, 2, 2 * 9 * 11, 10.0, None))
    assert (t[..., :2] - torch.tanh(x[..., :2]) * 10).abs().max() <= 1e-5
    big = torch.randn(16, 24, 8) * 2
    crop = torch.zeros(13, 20, 3)
    ok(lib.vfi_clamp_crop(P(big, 1), 8, 16, 24, P(crop), 13, 20, 3, None))
    assert torch.equal(crop, big[:13, :20, 1:4].clamp(0, 1))


def test_instance_norm_and_layer_norm(lib):
    torch.manual_seed(1)
    x = torch.randn(2, 20, 30, 40) * 3 + 1.5         # NHWC, 40-channel stride
    add = torch.randn(2, 20, 30, 32)
    stats = torch.zeros(2, 24, 2)
    ws = torch.zeros(2 * 64 * 24 * 2, dtype=torch.float64)
    ok(lib.vfi_instnorm_stats(P(x, 8), 40, 24, 2, 600, P(stats), ws.data_ptr(), ws.numel() * 8, None))
    xc = nchw(x[..., 8:32])
    want = F.instance_norm(xc)
    out = torch.zeros(2, 20, 30, 24)
    ok(lib.vfi_instnorm_apply(P(x, 8), 40, P(stats), 24, 2, 600, 0, None, 0, 0, P(out), 24, None))
    assert (nchw(out) - want).abs().max() <= 2e-6
    ok(lib.vfi_instnorm_apply(P(x, 8), 40, P(stats), 24, 2, 600, 1, P(add, 4), 32, 1, P(out), 24, None))
    assert (nchw(out) - F.relu(F.relu(want) + nchw(add[..., 4:28]))).abs().max() <= 2e-6
    tok = torch.randn(50, 136)
    gm, bt = torch.rand(128) + 0.5, torch.randn(128) * 0.1
    ln = torch.zeros(50, 128)
    ok(lib.vfi_layernorm(P(tok, 8), 136, 128, 50, P(gm), P(bt), P(ln), 128, None))
    assert (ln - F.layer_norm(tok[:, 8:], (128,), gm, bt)).abs().max() <= 2e-6
    # source + LN(message) in place on the source, second copy into a wider tensor's channel window (TransformerLayer :517-523)
    src, cat = torch.randn(50, 128), torch.full((50, 256), 7.0)
    want = src + F.layer_norm(tok[:, 8:], (128,), gm, bt)
    ok(lib.vfi_layernorm_add(P(tok, 8), 136, 128, 50, P(gm), P(bt), P(src), 128, P(src), 128, P(cat), 256, None))
    assert (src - want).abs().max() <= 2e-6 and torch.equal(cat[:, :128], src) and (cat[:, 128:] == 7.0).all()
    assert lib.vfi_layernorm_add(P(tok, 8), 136, 128, 50, P(gm), P(bt), P(src), 128, P(tok, 8), 136, None, 0, None) != 0   # out aliases x


@pytest.mark.parametrize("shifted", [False, True])
def test_window_attention_pipeline(lib, shifted):
    """partition -> q k^T / sqrt(c) (+ shift mask) -> softmax -> attn v -> merge, against the oracle's window attention"""
    torch.manual_seed(2)
    b, h, w, c, k = 2, 8, 12, 16, 2
    q, kk, v = (torch.randn(b, h * w, c) for _ in range(3))
    mask = G._shift_mask(h, w, h // k, w // k) if shifted else None
    want = G._window_attention(q, kk, v, k, shifted, h, w, mask)
    sh, sw = ((h // k) // 2, (w // k) // 2) if shifted else (0, 0)
    nb, lw = b * k * k, (h // k) * (w // k)
    qw, kw, vw = (torch.zeros(nb, lw, c) for _ in range(3))
    for src, dst in ((q, qw), (kk, kw), (v, vw)):
        ok(lib.vfi_window_partition(P(src), c, P(dst), c, b, h, w, c, k, sh, sw, 0, None))
    sc = torch.zeros(nb, lw, lw)
    ok(lib.vfi_bmm_nt(P(qw), c, P(kw), c, P(sc), nb, lw, lw, c, 1.0 / c ** 0.5, None))
    m = mask.contiguous() if shifted else None
    ok(lib.vfi_softmax_rows(P(sc), nb, lw, lw, P(m) if shifted else None, k * k, None))
    ow = torch.zeros(nb, lw, c)
    ok(lib.vfi_bmm_nn(P(sc), P(vw), c, P(ow), c, nb, lw, lw, c, None))
    out = torch.zeros(b, h * w, c)
    ok(lib.vfi_window_partition(P(ow), c, P(out), c, b, h, w, c, k, sh, sw, 1, None))
    assert (out - want).abs().max() <= 2e-6


def test_global_match_and_propagate(lib):
    torch.manual_seed(3)
    b, c, h, w = 1, 32, 6, 9
    f0, f1 = torch.randn(b, c, h, w), torch.randn(b, c, h, w)
    want = G.global_match(f0, f1)
    a0, a1 = nhwc(f0).view(b, h * w, c), nhwc(f1).view(b, h * w, c)
    sc = torch.zeros(b, h * w, h * w)
    ok(lib.vfi_bmm_nt(P(a0), c, P(a1), c, P(sc), b, h * w, h * w, c, 1.0 / c ** 0.5, None))
    ok(lib.vfi_softmax_rows(P(sc), b, h * w, h * w, None, 0, None))
    grid = nhwc(G._pixel_grid(b, h, w)).view(b, h * w, 2).contiguous()
    corr = torch.zeros(b, h * w, 2)
    ok(lib.vfi_bmm_nn(P(sc), P(grid), 2, P(corr), 2, b, h * w, h * w, 2, None))
    assert (nchw((corr - grid).view(b, h, w, 2)) - want).abs().max() <= 1e-5


def test_flow_sample_local_match_local_propagate(lib):
    torch.manual_seed(4)
    b, c, h, w = 2, 16, 10, 14
    f0, f1 = torch.randn(b, c, h, w), torch.randn(b, c, h, w)
    flow = torch.randn(b, 2, h, w) * 3
    got = torch.zeros(b, h, w, c)
    fl, a0, a1 = nhwc(flow), nhwc(f0), nhwc(f1)      # keep the NHWC copies alive while raw pointers are in use
    ok(lib.vfi_flow_sample(P(a1), c, P(fl), 2, P(got), c, b, h, w, c, None))
    assert (nchw(got) - G._flow_sample(f1, flow)).abs().max() <= 1e-5
    want = flow + G.local_match(f0, f1, 4)
    fl2 = nhwc(flow)
    ok(lib.vfi_local_match(P(a0), c, P(a1), c, P(fl2), 2, b, h, w, c, 4, None))
    assert (nchw(fl2) - want).abs().max() <= 2e-5
    sd = {"feature_flow_attn.q_proj.weight": torch.randn(c, c) * 0.3, "feature_flow_attn.q_proj.bias": torch.randn(c) * 0.1,
          "feature_flow_attn.k_proj.weight": torch.randn(c, c) * 0.3, "feature_flow_attn.k_proj.bias": torch.randn(c) * 0.1}
    want = G.propagate(sd, f0, flow, 1)
    tok = a0
    qp = F.linear(tok, sd["feature_flow_attn.q_proj.weight"], sd["feature_flow_attn.q_proj.bias"]).contiguous()
    kp = F.linear(tok, sd["feature_flow_attn.k_proj.weight"], sd["feature_flow_attn.k_proj.bias"]).contiguous()
    out = torch.zeros(b, h, w, 2)
    ok(lib.vfi_local_propagate(P(qp), c, P(kp), c, P(fl), 2, P(out), 2, b, h, w, c, 1, None))
    assert (nchw(out) - want).abs().max() <= 1e-5


def test_convex_upsample(lib):
    torch.manual_seed(5)
    b, h, w, k = 2, 5, 7, 4
    flow, logits = torch.randn(b, 2, h, w) * 2, torch.randn(b, 9 * k * k, h, w)
    mask = torch.softmax(logits.view(b, 1, 9, k, k, h, w), dim=2)
    up = F.unfold(k * flow, [3, 3], padding=1).view(b, 2, 9, 1, 1, h, w)
    want = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(b, 2, k * h, k * w)
    out = torch.zeros(b, k * h, k * w, 2)
    lg, fl = nhwc(logits), nhwc(flow)
    ok(lib.vfi_convex_upsample(P(lg), 9 * k * k, P(fl), 2, P(out), 2, b, h, w, k, None))
    assert (nchw(out) - want).abs().max() <= 1e-5


def test_metric_inputs(lib):
    torch.manual_seed(6)
    h, w = 12, 18
    i0, i1 = torch.rand(1, 3, h, w), torch.rand(1, 3, h, w)
    f01, f10 = torch.randn(1, 2, h, w) * 1.5, torch.randn(1, 2, h, w) * 1.5
    m0 = F.l1_loss(i0, G._backwarp_zeros(i1, f01), reduction="none").mean([1], True)
    m1 = F.l1_loss(i1, G._backwarp_zeros(i0, f10), reduction="none").mean([1], True)
    occ_f, occ_b = G._occlusion(f01, f10)
    n01 = torch.cat([f01[:, 0:1] / ((w - 1.0) / 2.0), f01[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    n10 = torch.cat([f10[:, 0:1] / ((w - 1.0) / 2.0), f10[:, 1:2] / ((h - 1.0) / 2.0)], 1)
    want = nhwc(torch.cat((i0, i1, -m0, -m1, n01, n10, occ_f.unsqueeze(1), occ_b.unsqueeze(1)), 1))
    a0, a1 = torch.zeros(h, w, 8), torch.zeros(h, w, 8)
    a0[..., :3], a1[..., :3] = nhwc(i0)[0], nhwc(i1)[0]
    out = torch.zeros(h, w, 16)
    fa, fb = nhwc(f01), nhwc(f10)
    ok(lib.vfi_gmfss_metric_inputs(P(a0), P(a1), 8, P(fa), P(fb), 2, P(out), 16, h, w, None))
    d = (out[..., :14] - want[0]).abs()
    assert d[..., :12].max() <= 1e-5
    assert (d[..., 12:] > 0).float().mean() <= 0.01          # the occlusion bits are thresholds: allow rare boundary flips


def test_splat_wrapper_and_pixel_shuffle(lib):
    torch.manual_seed(7)
    n, c, h, w = 1, 5, 9, 13
    x, z, flow = torch.randn(n, c, h, w), torch.randn(n, 1, h, w), torch.randn(n, 2, h, w) * 2
    t = 0.3
    pre = torch.zeros(n * h * w, c + 1)
    fo = torch.zeros(n * h * w, 2)
    xs = torch.zeros(n, h, w, 8)
    xs[..., 1:6] = nhwc(x)
    zz, ff = nhwc(z), nhwc(flow)
    ok(lib.vfi_splat_prep(P(xs, 1), 8, P(zz), 1, P(ff), 2, P(pre), P(fo), c, n * h * w, t, t, None))
    zt = t * z
    want_pre = nhwc(torch.cat([x * zt.exp(), zt.exp()], 1)).view(-1, c + 1)
    assert (pre - want_pre).abs().max() <= 1e-5 and torch.equal(fo, nhwc(t * flow).view(-1, 2))
    s = G.splat_sum(torch.cat([x * zt.exp(), zt.exp()], 1), t * flow)
    want = s[:, :-1] / (s[:, -1:] + 0.0000001)
    sn = nhwc(s).view(-1, c + 1).contiguous()
    out = torch.zeros(n, h, w, 8)
    ok(lib.vfi_splat_normalize(P(sn), P(out, 2), 8, c, n * h * w, None))
    assert (nchw(out[..., 2:7]) - want).abs().max() <= 1e-5
    y = torch.randn(2, 4 * 6, 5, 7)
    ps = torch.zeros(2, 10, 14, 8)
    yy = nhwc(y)
    ok(lib.vfi_pixel_shuffle2(P(yy), 24, P(ps, 1), 8, 2, 5, 7, 6, None))
    assert torch.equal(nchw(ps[..., 1:7]), F.pixel_shuffle(y, 2))


def test_resize_align_corners(lib):
    torch.manual_seed(8)
    x = torch.randn(2, 2, 9, 15)
    want = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) * 2
    xi = torch.zeros(2, 9, 15, 4)
    xi[..., 1:3] = nhwc(x)
    out = torch.zeros(2, 18, 30, 2)
    ok(lib.vfi_resize_bilinear_ac(P(xi, 1), 4, P(out), 2, 2, 9, 15, 18, 30, 2, 2.0, None))
    assert (nchw(out) - want).abs().max() <= 1e-5


@pytest.mark.parametrize("nb,m,n,k,acs", [(3, 10, 13, 16, 16), (2, 8, 8, 128, 136), (1, 5, 3, 6, 6), (2, 7, 9, 12, 14)])
def test_bmm_nt_tiled_and_scalar_paths(lib, nb, m, n, k, acs):
    """K % 4 == 0 with aligned rows takes the 4x4-tile body (edge tiles: M, N not multiples of 4), otherwise the scalar body"""
    torch.manual_seed(9)
    a, b = torch.randn(nb, m, acs), torch.randn(nb, n, acs)
    out = torch.full((nb, m, n), 7.0)
    ok(lib.vfi_bmm_nt(P(a), acs, P(b), acs, P(out), nb, m, n, k, 0.5, None))
    assert (out - 0.5 * torch.matmul(a[..., :k], b[..., :k].transpose(1, 2))).abs().max() <= 1e-5
