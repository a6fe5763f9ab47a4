"""CPU: the per-element bodies of csrc/ifunet_bodies.h (the code the MI355X kernels of csrc/ifunet_ops.hip execute), run on the
host through tests/hostcheck and compared with the torch expression of the reference they replace
(vfi_models/ifunet/IFUNet_arch.py; restated in oracle/ifunet_oracle.py)."""
import pytest
import torch
import torch.nn.functional as F

import hostcheck
from oracle import ifunet_oracle as O


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


def test_cbam_pipeline(lib):
    """channel pool -> gate MLP -> scale + compress -> 7x7 spatial gate, against oracle.cbam on the same parameters"""
    torch.manual_seed(0)
    n, c, h, w, r = 2, 64, 9, 13, 4
    sd = {"m.ChannelGate.mlp.1.weight": torch.randn(r, c) * 0.2, "m.ChannelGate.mlp.1.bias": torch.randn(r) * 0.1,
          "m.ChannelGate.mlp.3.weight": torch.randn(c, r) * 0.2, "m.ChannelGate.mlp.3.bias": torch.randn(c) * 0.1,
          "m.SpatialGate.spatial.conv.weight": torch.randn(1, 2, 7, 7) * 0.1, "m.SpatialGate.spatial.bn.weight": torch.tensor([1.2]),
          "m.SpatialGate.spatial.bn.bias": torch.tensor([-0.1]), "m.SpatialGate.spatial.bn.running_mean": torch.tensor([0.05]),
          "m.SpatialGate.spatial.bn.running_var": torch.tensor([0.8])}
    x = torch.randn(n, c, h, w)
    want = O.cbam(sd, "m", x)
    xi = torch.zeros(n, h, w, c + 8)
    xi[..., 8:] = nhwc(x)
    stats = torch.zeros(n, c, 2)
    ws = torch.zeros(n * 64 * c * 3, dtype=torch.float32)
    ok(lib.vfi_channel_pool(P(xi, 8), c + 8, c, n, h * w, P(stats), ws.data_ptr(), ws.numel() * 4, None))
    assert (stats[..., 0] - x.mean((2, 3))).abs().max() <= 1e-6 and torch.equal(stats[..., 1], x.amax((2, 3)))
    scale = torch.zeros(n, c)
    w1, b1, w2, b2 = (sd["m.ChannelGate.mlp." + k].contiguous() for k in ("1.weight", "1.bias", "3.weight", "3.bias"))
    ok(lib.vfi_cbam_gate(P(stats), P(w1), P(b1), P(w2), P(b2), c, r, n, P(scale), None))
    xs, comp = torch.zeros(n, h, w, c), torch.zeros(n * h * w, 2)
    ok(lib.vfi_cbam_scale_compress(P(xi, 8), c + 8, P(scale), c, n, h * w, P(xs), c, P(comp), None))
    a = float(sd["m.SpatialGate.spatial.bn.weight"] / torch.sqrt(sd["m.SpatialGate.spatial.bn.running_var"] + 1e-5))
    bsh = float(sd["m.SpatialGate.spatial.bn.bias"] - sd["m.SpatialGate.spatial.bn.running_mean"] * a)
    w7 = sd["m.SpatialGate.spatial.conv.weight"][0].permute(1, 2, 0).contiguous()      # [7][7][2]
    ok(lib.vfi_cbam_spatial(P(xs), c, P(comp), P(w7), a, bsh, c, n, h, w, None))
    assert (nchw(xs) - want).abs().max() <= 2e-6


@pytest.mark.parametrize("k,fc", [(4, 4), (16, 4), (8, 2)])
def test_convex_upsample_channels(lib, k, fc):
    torch.manual_seed(1)
    n, h, w = 2, 3, 5
    flow, logits = torch.randn(n, fc, h, w) * 2, torch.randn(n, 9 * k * k, h, w)
    mask = torch.softmax(logits.view(n, 1, 9, k, k, h, w), dim=2)
    up = F.unfold(k * flow, [3, 3], padding=1).view(n, fc, 9, 1, 1, h, w)
    want = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3).reshape(n, fc, k * h, k * w)
    lg, fl = nhwc(logits), nhwc(flow)
    out = torch.zeros(n, k * h, k * w, 8)
    ok(lib.vfi_convex_upsample_c(P(lg), 9 * k * k, P(fl), fc, P(out, 2), 8, n, h, w, k, fc, None))
    assert (nchw(out[..., 2:2 + fc]) - want).abs().max() <= 1e-5


def test_blends_fill_clamp(lib):
    torch.manual_seed(2)
    hp, wp, h, w = 16, 24, 13, 20
    a, b, d = (torch.rand(1, hp, wp, 4) for _ in range(3))
    m = torch.rand(1, hp, wp, 1)
    out = torch.zeros(1, hp, wp, 4)
    ok(lib.vfi_lerp_mask(P(a), 4, P(b), 4, P(m), 1, P(out), 4, 3, hp * wp, None))
    assert (out[..., :3] - (a[..., :3] * m + b[..., :3] * (1 - m))).abs().max() <= 1e-6
    x, y = torch.randn(1, hp, wp, 4), torch.randn(1, hp, wp, 4)
    ok(lib.vfi_add_clamp01(P(x), 4, P(y), 4, P(out), 4, 3, hp * wp, None))
    assert torch.equal(out[..., :3], (x[..., :3] + y[..., :3]).clamp(0, 1))
    m0, m1 = torch.randn(1, hp, wp, 1) * 5, torch.randn(1, hp, wp, 1) * 5       # beyond +-4: the clamp matters
    res = torch.zeros(h, w, 3)
    ok(lib.vfi_ifunet_blend(P(a), P(b), P(d), 4, P(m0), P(m1), 1, P(res), hp, wp, h, w, None))
    mk = F.softmax(torch.clamp(torch.cat((m0, m1, m1 * 0), 3), -4, 4), dim=3)
    want = (a[..., :3] * mk[..., 0:1] + b[..., :3] * mk[..., 1:2] + d[..., :3] * mk[..., 2:3])[0, :h, :w]
    assert (res - want).abs().max() <= 1e-6
    t = torch.zeros(2, 5, 6, 8)
    ok(lib.vfi_fill_channels(P(t, 3), 8, 2, 60, 0.75, None))
    assert (t[..., 3:5] == 0.75).all() and t[..., :3].abs().max() == 0 and t[..., 5:].abs().max() == 0
