"""-m gpu: the device forms of GMFlow's steps that do not run a per-element body of csrc/gmfss_bodies.h on the MI355X
(csrc/gmfss_fast.hip, the window mode of csrc/attention.hip), against the torch statement of the reference code they replace
(vfi_models/gmfss_fortuna/GMFSS_Fortuna_union_arch.py, restated in oracle/gmfss_oracle.py):
  * vfi_layernorm        one wave per token                                   (:479-523, nn.LayerNorm)
  * vfi_local_match      fp32-MFMA banded products + the body's tap arithmetic (:846-913, local_correlation_softmax)
  * vfi_window_attention roll / split / attention / merge / roll back in one kernel (:367-436)
The bodies themselves are checked on the host by tests/test_gmfss_bodies_cpu.py; these tests make sure the dispatch on the
device computes the same thing, including image sizes that are not multiples of the kernels' tiles."""
import pytest
import torch
import torch.nn.functional as F

from gpu_util import describe_diff
from oracle import gmfss_oracle as G

pytestmark = pytest.mark.gpu


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("tokens,c,cs,ocs", [(1000, 128, 128, 128), (4099, 128, 256, 136), (37, 96, 96, 96), (300, 256, 256, 256), (50, 320, 320, 320)])
def test_layernorm(hip_lib, tokens, c, cs, ocs):
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(tokens + c)
    x = torch.randn(tokens, cs, generator=g) * 3 + 1
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    want = F.layer_norm(x[:, :c].double(), (c,), gamma.double(), beta.double(), 1e-5).float()
    xd, gd, bd = x.cuda(), gamma.cuda(), beta.cuda()
    out = torch.full((tokens, ocs), float("nan"), device="cuda")
    _lib.check(hip_lib.vfi_layernorm(xd.data_ptr(), cs, c, tokens, gd.data_ptr(), bd.data_ptr(), out.data_ptr(), ocs, None), "vfi_layernorm")
    torch.cuda.synchronize()
    got = out.cpu()
    assert (got[:, :c] - want).abs().max().item() <= 2e-5, describe_diff(got[:, :c], want, "layernorm", chan_last=False)
    assert ocs == c or torch.isnan(got[:, c:]).all(), "wrote outside its channel window"
    # vfi_layernorm_add: out = add + LN(x), in place on add, plus the second copy into a wider tensor
    add = torch.randn(tokens, c, generator=g)
    addd = add.cuda()
    cat = torch.full((tokens, 2 * c), float("nan"), device="cuda")
    _lib.check(hip_lib.vfi_layernorm_add(xd.data_ptr(), cs, c, tokens, gd.data_ptr(), bd.data_ptr(), addd.data_ptr(), c, addd.data_ptr(), c,
                                         cat.data_ptr(), 2 * c, None), "vfi_layernorm_add")
    torch.cuda.synchronize()
    assert (addd.cpu() - (add + want)).abs().max().item() <= 2e-5
    assert torch.equal(cat[:, :c].cpu(), addd.cpu()) and torch.isnan(cat[:, c:]).all()


@pytest.mark.parametrize("b,h,w,gain", [(1, 16, 24, 1.0), (2, 13, 21, 1.0), (1, 34, 60, 3.0), (1, 136, 240, 1.0), (1, 5, 9, 1.0)])
def test_local_match_mfma(hip_lib, b, h, w, gain):
    """C = 128, radius 4: GMFlow's refinement step — the shape that takes the matrix-core kernel"""
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(h * 100 + w)
    f0 = torch.randn(b, 128, h, w, generator=g) * gain
    f1 = torch.randn(b, 128, h, w, generator=g) * gain
    flow = torch.randn(b, 2, h, w, generator=g) * 3
    want = flow + G.local_match(f0, f1, 4)
    a0, a1, fl = nhwc(f0).cuda(), nhwc(f1).cuda(), nhwc(flow).cuda()
    _lib.check(hip_lib.vfi_local_match(a0.data_ptr(), 128, a1.data_ptr(), 128, fl.data_ptr(), 2, b, h, w, 128, 4, None), "vfi_local_match")
    torch.cuda.synchronize()
    got = fl.cpu().permute(0, 3, 1, 2)
    # logits grow with gain^2 (|q . k| / sqrt(C) ~ 100 at gain 3): their fp32 rounding is what the softmax amplifies, in torch too.
    # Two terms grow with the image: the reference forms  sum p (X + dx) - X  in fp32 (X * 2^-23 per term; the kernel sums p dx),
    # and its window positions go through grid_sample's normalise / un-normalise round trip (off an integer by ~ W * 2^-23 px,
    # reproduced by the kernel as a blend with the neighbouring pixel).  Checked against the fp32 oracle and against the exact
    # (float64, integer-position) value, which the oracle itself misses by 1.1e-4 at w = 240 and by 1.5e-4 at gain 3.
    tol = 3e-5 * gain ** 2 + 1e-6 * max(h, w)
    exact = flow + _local_match_f64(f0, f1, 4)
    assert (got - exact).abs().max().item() <= tol, describe_diff(got, exact, f"local_match vs f64 {b}x{h}x{w} gain {gain}")
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"local_match {b}x{h}x{w} gain {gain}")


def _local_match_f64(f0, f1, r):
    """local_correlation_softmax (:846-913) in float64 with the window read at integer positions (what the grid_sample round
    trip of the reference is up to 1e-5 px): expected OFFSET of softmax(q . k / sqrt(C)) over the in-image window positions"""
    b, c, h, w = f0.shape
    q, k = f0.double(), F.pad(f1.double(), (r, r, r, r))
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    logits, offs = [], []
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            s = (q * k[:, :, r + dy:r + dy + h, r + dx:r + dx + w]).sum(1) / c ** 0.5
            valid = (ys + dy >= 0) & (ys + dy < h) & (xs + dx >= 0) & (xs + dx < w)
            logits.append(torch.where(valid[None], s, torch.tensor(-1e9, dtype=torch.float64)))
            offs.append((dx, dy))
    p = torch.softmax(torch.stack(logits, 1), dim=1)                      # [b, 81, h, w]
    o = torch.tensor(offs, dtype=torch.float64)                           # [81, 2] = (dx, dy)
    return torch.einsum("bjhw,jc->bchw", p, o).float()


def test_local_match_body_path(hip_lib):
    """any other channel count / radius still runs the per-element body"""
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(9)
    f0, f1 = torch.randn(1, 16, 10, 14, generator=g), torch.randn(1, 16, 10, 14, generator=g)
    flow = torch.randn(1, 2, 10, 14, generator=g)
    want = flow + G.local_match(f0, f1, 3)
    a0, a1, fl = nhwc(f0).cuda(), nhwc(f1).cuda(), nhwc(flow).cuda()
    _lib.check(hip_lib.vfi_local_match(a0.data_ptr(), 16, a1.data_ptr(), 16, fl.data_ptr(), 2, 1, 10, 14, 16, 3, None), "vfi_local_match")
    torch.cuda.synchronize()
    assert (fl.cpu().permute(0, 3, 1, 2) - want).abs().max().item() <= 2e-5


@pytest.mark.parametrize("B,h,w,splits,shifted", [(2, 16, 24, 2, False), (2, 16, 24, 2, True), (4, 34, 60, 2, True), (2, 136, 240, 8, True),
                                                  (1, 136, 240, 8, False)])
def test_window_attention(hip_lib, B, h, w, splits, shifted):
    from cfi_amd import _lib
    from cfi_amd.gmfss import shift_labels

    g = torch.Generator().manual_seed(B * 1000 + h + splits + int(shifted))
    c = 128
    q, k, v = (torch.randn(B, h, w, c, generator=g) for _ in range(3))
    wh, ww = h // splits, w // splits
    sh, sw = (wh // 2, ww // 2) if shifted else (0, 0)
    labels = shift_labels(h, w, splits) if shifted else None

    def windows(t):
        t = torch.roll(t, shifts=(-sh, -sw), dims=(1, 2))
        return t.view(B, splits, wh, splits, ww, c).permute(0, 1, 3, 2, 4, 5).reshape(B * splits * splits, wh * ww, c).double()

    sc = torch.matmul(windows(q), windows(k).transpose(1, 2)) / c ** 0.5
    if shifted:
        lab = labels.view(splits * splits, wh * ww)
        sc = sc + ((lab[:, :, None] != lab[:, None, :]).double() * -100.0).repeat(B, 1, 1)
    o = torch.matmul(torch.softmax(sc, dim=-1), windows(v)).float()
    o = o.view(B, splits, splits, wh, ww, c).permute(0, 1, 3, 2, 4, 5).reshape(B, h, w, c)
    want = torch.roll(o, shifts=(sh, sw), dims=(1, 2))
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    ld = labels.cuda() if shifted else None
    out = torch.full((B, h, w, c), float("nan"), device="cuda")
    _lib.check(hip_lib.vfi_window_attention(qd.data_ptr(), c, kd.data_ptr(), c, vd.data_ptr(), c, out.data_ptr(), c, B, h, w, splits, sh, sw, c,
                                            1.0 / c ** 0.5, ld.data_ptr() if shifted else None, None), "vfi_window_attention")
    torch.cuda.synchronize()
    got = out.cpu()
    assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), describe_diff(got, want, "window attention")
    with pytest.raises(RuntimeError, match="aliases"):
        _lib.check(hip_lib.vfi_window_attention(qd.data_ptr(), c, kd.data_ptr(), c, vd.data_ptr(), c, qd.data_ptr(), c, B, h, w, splits, sh, sw, c,
                                                1.0, None, None), "vfi_window_attention")


@pytest.mark.parametrize("b,h,w", [(1, 10, 14), (2, 33, 47), (1, 136, 240)])
def test_local_propagate_cooperative(hip_lib, b, h, w):
    """C = 128, radius 1 (GMFlow's local flow propagation): 16 lanes per pixel instead of the per-pixel body"""
    from cfi_amd import _lib

    g = torch.Generator().manual_seed(h + w)
    c = 128
    f0 = torch.randn(b, c, h, w, generator=g)
    flow = torch.randn(b, 2, h, w, generator=g) * 3
    sd = {"feature_flow_attn.q_proj.weight": torch.randn(c, c, generator=g) * 0.1, "feature_flow_attn.q_proj.bias": torch.randn(c, generator=g) * 0.1,
          "feature_flow_attn.k_proj.weight": torch.randn(c, c, generator=g) * 0.1, "feature_flow_attn.k_proj.bias": torch.randn(c, generator=g) * 0.1}
    want = G.propagate(sd, f0, flow, 1)
    tok = nhwc(f0)
    qp = F.linear(tok, sd["feature_flow_attn.q_proj.weight"], sd["feature_flow_attn.q_proj.bias"]).contiguous().cuda()
    kp = F.linear(tok, sd["feature_flow_attn.k_proj.weight"], sd["feature_flow_attn.k_proj.bias"]).contiguous().cuda()
    fl = nhwc(flow).cuda()
    out = torch.full((b, h, w, 2), float("nan"), device="cuda")
    _lib.check(hip_lib.vfi_local_propagate(qp.data_ptr(), c, kp.data_ptr(), c, fl.data_ptr(), 2, out.data_ptr(), 2, b, h, w, c, 1, None), "vfi_local_propagate")
    torch.cuda.synchronize()
    got = out.cpu().permute(0, 3, 1, 2)
    assert (got - want).abs().max().item() <= 2e-5, describe_diff(got, want, f"local_propagate {b}x{h}x{w}")
