"""-m gpu: M2M network path on the MI355X vs the oracle (oracle/m2m_model_oracle.py, bit-exact vs the reference's
M2M_arch in the build container with the C restatements of the two cupy ops).  Layer objects and M2M-specific
kernels first, then the whole model, then the node loop.  Tolerance: per-pixel fp32 |d| <= 1e-3 (BASELINE.json)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from gpu_util import describe_diff, nhwc
from cfi_amd import synth
from cfi_amd.schedule import InterpolationStateList
from oracle import m2m_model_oracle as mo

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def lib(hip_lib):
    from cfi_amd import _lib

    _lib.check(hip_lib.vfi_init(0), "vfi_init")
    return hip_lib


@pytest.fixture(scope="module")
def sd():
    return synth.m2m_synth_state_dict(1234)


@pytest.fixture(scope="module")
def engine(lib, sd):
    from cfi_amd.m2m import M2MEngine

    e = M2MEngine(sd)
    yield e
    e.close()


def _ck(rc, what):
    from cfi_amd import _lib

    _lib.check(rc, what)


def _rand(g, *shape):
    return torch.rand(*shape, generator=g) * 2 - 1


# kind, k, stride, pad_mode, act, cin, cout, n, h, w
LAYERS = [
    (0, 2, 2, 0, 1, 3, 32, 2, 24, 40),        # sconv(2)-prelu (extractor, first stage)
    (0, 2, 2, 0, 1, 32, 32, 2, 16, 24),
    (0, 3, 1, 1, 1, 115, 128, 2, 13, 21),     # decoder conv(3, replpad)-prelu
    (0, 3, 1, 1, 1, 128, 96, 1, 9, 17),
    (0, 3, 1, 1, 0, 32, 2, 2, 11, 19),        # decoder flow head (with residual, see below)
    (0, 3, 2, 0, 3, 8, 32, 2, 32, 48),        # Conv2.conv1: stride 2, zero pad, per-channel PReLU
    (0, 3, 2, 0, 3, 96, 64, 2, 20, 28),
    (0, 3, 2, 0, 3, 384, 256, 2, 8, 12),
    (0, 3, 1, 0, 3, 256, 256, 2, 5, 7),
    (0, 3, 1, 0, 0, 16, 9, 2, 21, 30),        # flow-residual + mask-logit head
    (0, 1, 1, 0, 4, 256, 4096, 2, 1, 1),      # cube conv_C (sigmoid)
    (0, 1, 1, 0, 4, 256, 16, 2, 9, 1),        # cube conv_H
    (0, 3, 1, 0, 1, 1920, 256, 1, 16, 30),    # FILM's coarsest flow-estimator conv: 32 workgroups -> split-K over 16 (conv_mfma2.hip)
    (0, 3, 1, 0, 1, 1920, 256, 1, 33, 60),    # ... split over fewer
    (0, 1, 1, 0, 5, 1024, 128, 1, 17, 30),    # 1x1 + GELU, split-K: the epilogue runs in the reduce kernel
    (0, 3, 2, 0, 3, 768, 128, 1, 18, 30),     # stride 2 + per-channel PReLU, split-K
    (0, 1, 1, 0, 5, 256, 1024, 2, 34, 60),    # GMFlow's FFN: Linear(2c, 8c) + nn.GELU() in the epilogue (interior tiles)
    (0, 1, 1, 0, 5, 256, 1024, 1, 5, 7),      # ... and border tiles
    (0, 3, 1, 0, 5, 16, 9, 1, 9, 11),         # GELU through the generic epilogue
    (1, 4, 2, 0, 3, 768, 128, 2, 4, 6),       # deconv(): ConvTranspose2d(4, 2, 1) + PReLU
    (1, 4, 2, 0, 3, 64, 16, 2, 12, 20),
]


@pytest.mark.parametrize("kind,k,stride,pad_mode,act,cin,cout,n,h,w", LAYERS)
def test_layer(lib, kind, k, stride, pad_mode, act, cin, cout, n, h, w):
    g = torch.Generator().manual_seed(cin * 31 + cout + k)
    x = _rand(g, n, cin, h, w)
    wt = _rand(g, *((cout, cin, k, k) if kind == 0 else (cin, cout, k, k))) / (cin * k * k / (4 if kind else 1)) ** 0.5
    b = _rand(g, cout) * 0.5
    pre = 0.1 + 0.3 * torch.rand(cout, generator=g)
    slope = 0.17
    use_res = cout == 2
    if kind == 1:
        y = F.conv_transpose2d(x, wt, b, 2, 1)
    elif k == 3 and pad_mode == 1:
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), wt, b, stride)
    else:
        y = F.conv2d(x, wt, b, stride, 1 if k == 3 else 0)
    res = _rand(g, *y.shape) if use_res else None
    if use_res:
        y = y + res
    y = {0: lambda v: v, 1: lambda v: F.leaky_relu(v, slope), 3: lambda v: F.prelu(v, pre), 4: torch.sigmoid, 5: F.gelu}[act](y)
    want = nhwc(y)
    cphys = (cin + 7) // 8 * 8
    off_in, off_out = 8, 3
    xin = torch.zeros(n, h, w, cphys + 16)
    xin[..., off_in:off_in + cin] = nhwc(x)
    xd = xin.cuda()
    out = torch.full((n,) + tuple(want.shape[1:3]) + (cout + 5,), float("nan"), device="cuda")
    resd = None
    if use_res:
        resd = torch.zeros(n, want.shape[1], want.shape[2], 6)
        resd[..., 2:4] = nhwc(res)
        resd = resd.cuda()
    hnd = lib.vfi_conv_create_ex(kind, wt.contiguous().data_ptr(), b.data_ptr(), cout, cin, k, stride, pad_mode, None, cphys,
                                 pre.data_ptr() if act == 3 else None)
    assert hnd
    try:
        _ck(lib.vfi_conv_forward_ex(hnd, xd.data_ptr() + 4 * off_in, xd.shape[-1], h, w, out.data_ptr() + 4 * off_out, out.shape[-1], n,
                                    act, slope, 0.0, 0.0, resd.data_ptr() + 8 if use_res else None, 6 if use_res else 0, None), "fwd")
        torch.cuda.synchronize()
    finally:
        lib.vfi_conv_destroy(hnd)
    got = out.cpu()
    assert torch.isnan(got[..., :off_out]).all() and torch.isnan(got[..., off_out + cout:]).all(), "wrote outside the window"
    got = got[..., off_out:off_out + cout]
    assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), describe_diff(got, want, "layer")


def test_layer_post_affine(lib):
    g = torch.Generator().manual_seed(5)
    x, wt, b = _rand(g, 1, 16, 12, 20), _rand(g, 1, 16, 3, 3) * 0.2, _rand(g, 1)
    want = nhwc(torch.sigmoid(F.conv2d(x, wt, b, 1, 1)) * 0.8 + 0.1)
    xd = nhwc(x).cuda()
    out = torch.zeros(1, 12, 20, 1, device="cuda")
    hnd = lib.vfi_conv_create_ex(0, wt.data_ptr(), b.data_ptr(), 1, 16, 3, 1, 0, None, 16, None)
    _ck(lib.vfi_conv_forward_ex(hnd, xd.data_ptr(), 16, 12, 20, out.data_ptr(), 1, 1, 4, 0.0, 0.8, 0.1, None, 0, None), "fwd")
    torch.cuda.synchronize()
    lib.vfi_conv_destroy(hnd)
    assert (out.cpu() - want).abs().max().item() <= 1e-5


def test_layer_rejects(lib):
    from cfi_amd import _lib

    w = torch.zeros(4, 4, 5, 5)
    assert not lib.vfi_conv_create_ex(0, w.data_ptr(), None, 4, 4, 5, 1, 0, None, 8, None)
    assert "unsupported" in _lib.last_error()
    w = torch.zeros(32, 8, 3, 3)
    hnd = lib.vfi_conv_create_ex(0, w.data_ptr(), None, 32, 8, 3, 2, 0, None, 8, None)
    x = torch.zeros(1, 5, 6, 8, device="cuda")
    o = torch.zeros(1, 2, 3, 32, device="cuda")
    assert lib.vfi_conv_forward_ex(hnd, x.data_ptr(), 8, 5, 6, o.data_ptr(), 32, 1, 0, 0.0, 0.0, 0.0, None, 0, None) != 0   # odd size
    assert lib.vfi_conv_forward_ex(hnd, x.data_ptr(), 8, 4, 6, o.data_ptr(), 32, 1, 3, 0.0, 0.0, 0.0, None, 0, None) != 0   # no slopes
    lib.vfi_conv_destroy(hnd)


@pytest.mark.parametrize("n,c,h,w,amp", [(2, 3, 20, 31, 3.0), (2, 48, 17, 17, 2.0), (2, 32, 9, 14, 30.0), (4, 8, 6, 5, 1.0)])
def test_warp_m2m(lib, n, c, h, w, amp):
    g = torch.Generator().manual_seed(c * 3 + h)
    x = _rand(g, n, c, h, w)
    fl = _rand(g, n, 2, h, w) * amp   # amp 30 on a 9x14 image: most taps leave the image (zeros padding)
    want = nhwc(mo.backwarp(torch.stack([x[i ^ 1] for i in range(n)]), fl))   # in_swap: sample the partner image
    cs_in = c + (5 if c % 4 else 8)
    xin = torch.zeros(n, h, w, cs_in)
    ioff = 2 if c % 4 else 4
    xin[..., ioff:ioff + c] = nhwc(x)
    xd, fd = xin.cuda(), nhwc(fl).contiguous().cuda()
    out = torch.full((n, h, w, c + 4), float("nan"), device="cuda")
    _ck(lib.vfi_warp_m2m(xd.data_ptr() + 4 * ioff, cs_in, 1, fd.data_ptr(), 2, out.data_ptr(), c + 4, n, h, w, c, None), "warp")
    torch.cuda.synchronize()
    got = out.cpu()[..., :c]
    assert (got - want).abs().max().item() <= 1e-5, describe_diff(got, want, "warp_m2m")
    assert torch.isnan(out.cpu()[..., c:]).all()


@pytest.mark.parametrize("h,w,c", [(50, 70, 3), (64, 64, 4)])
def test_normalize(lib, h, w, c):
    g = torch.Generator().manual_seed(h)
    f0, f1 = torch.rand(h, w, c, generator=g), torch.rand(h, w, c, generator=g) * 0.6 + 0.3
    hp, wp = (h + 63) // 64 * 64, (w + 63) // 64 * 64
    ims = [F.pad(f[..., :3].permute(2, 0, 1)[None], [0, wp - w, 0, hp - h], mode="replicate") for f in (f0, f1)]
    mean_ = sum(t.mean([1, 2, 3], True) for t in ims) / 2
    std_ = (sum(t.std([1, 2, 3], False, True).square() + (mean_ - t.mean([1, 2, 3], True)).square() for t in ims) / 2).sqrt()
    want = torch.cat([nhwc((t - mean_) / (std_ + 0.0000001)) for t in ims])
    out = torch.zeros(2, hp, wp, 8, device="cuda")
    stats = torch.zeros(2, device="cuda")
    ws = torch.zeros(16384, dtype=torch.uint8, device="cuda")
    a, b = f0.cuda(), f1.cuda()
    _ck(lib.vfi_m2m_normalize(a.data_ptr(), b.data_ptr(), c, h, w, hp, wp, out.data_ptr(), 8, 2, stats.data_ptr(), ws.data_ptr(), 16384,
                              None), "normalize")
    torch.cuda.synchronize()
    st = stats.cpu()
    assert abs(st[0].item() - mean_.item()) <= 1e-6 and abs(st[1].item() - (std_.item() + 1e-7)) <= 1e-6
    got = out.cpu()
    assert (got[..., 2:5] - want).abs().max().item() <= 2e-5, describe_diff(got[..., 2:5], want, "normalize")
    assert (got[..., :2] == 0).all() and (got[..., 5:] == 0).all()


def test_cube(lib, sd):
    """pools + the three 1x1 sigmoid convs + cube_apply vs oracle._cube (EncDec attention, M2M_arch.py:786-795)"""
    class _Layer:     # a 1x1 conv of the checkpoint as a vfi_conv layer object, called with an explicit activation
        def __init__(self, lib_, w, b):
            w, b = w.detach().float().contiguous(), b.detach().float().contiguous()
            self.lib, self.keep = lib_, (w, b)
            self.h = lib_.vfi_conv_create_ex(0, w.data_ptr(), b.data_ptr(), w.shape[0], w.shape[1], 1, 1, 0, None, (w.shape[1] + 7) // 8 * 8, None)
            assert self.h

        def __call__(self, src, soff, dst, doff, act):
            n, hin, win, cs = src.shape
            _ck(self.lib.vfi_conv_forward_ex(self.h, src.data_ptr() + 4 * soff, cs, hin, win, dst.data_ptr() + 4 * doff, dst.shape[-1], n, act,
                                             0.0, 0.0, 0.0, None, 0, None), "conv_forward_ex")

        def close(self):
            self.lib.vfi_conv_destroy(self.h)

    g = torch.Generator().manual_seed(11)
    h, w = 7, 10
    s3 = _rand(g, 2, 256, h, w)
    want = nhwc(mo._cube(sd, s3))
    q = "MRN.motion_encdec."
    ls = [_Layer(lib, sd[q + f"conv_{n}.1.weight"], sd[q + f"conv_{n}.1.bias"]) for n in ("C", "H", "W")]
    z = lambda *s: torch.zeros(*s, device="cuda")
    sd3 = nhwc(s3).cuda()
    pc, ph, pw = z(2, 1, 1, 256), z(2, h, 1, 256), z(2, 1, w, 256)
    cc, ch, cw = z(2, 1, 1, 4096), z(2, h, 1, 16), z(2, 1, w, 16)
    for mode, dst in ((0, pc), (1, ph), (2, pw)):
        _ck(lib.vfi_pool_mean(sd3.data_ptr(), 256, dst.data_ptr(), 256, 2, h, w, 256, mode, None), "pool")
    torch.cuda.synchronize()
    assert (pc.cpu()[:, 0, 0] - s3.mean((2, 3))).abs().max().item() <= 1e-6
    assert (ph.cpu()[:, :, 0] - s3.mean(3).permute(0, 2, 1)).abs().max().item() <= 1e-6
    assert (pw.cpu()[:, 0] - s3.mean(2).permute(0, 2, 1)).abs().max().item() <= 1e-6
    ls[0](pc, 0, cc, 0, 4)
    ls[1](ph, 0, ch, 0, 4)
    ls[2](pw, 0, cw, 0, 4)
    out = z(2, h, w, 300)
    _ck(lib.vfi_m2m_cube_apply(sd3.data_ptr(), 256, cc.data_ptr(), ch.data_ptr(), 16, cw.data_ptr(), 16, out.data_ptr() + 16, 300, 2, h, w,
                               256, None), "cube")
    torch.cuda.synchronize()
    for l in ls:
        l.close()
    got = out.cpu()[..., 4:260]
    assert (got - want).abs().max().item() <= 1e-5, describe_diff(got, want, "cube")


def _frames(n, h, w, seed=3, shift=4.0):
    return synth.smooth_frames(n, h, w, seed=seed, shift=shift)


def _oracle_mid(sd, fr, ts):
    x = fr[..., :3].permute(0, 3, 1, 2)
    with torch.inference_mode():
        outs, aux = mo.m2m_forward(sd, x[0:1], x[1:2], [torch.tensor([t]).view(1, 1, 1, 1) for t in ts], return_aux=True)
    return [nhwc(o)[0] for o in outs], aux


@pytest.mark.parametrize("h,w", [(100, 150), (128, 128), (65, 200), (64, 64), (20, 30)])
def test_m2m_forward(engine, sd, h, w):
    fr = _frames(2, h, w)
    ts = [0.5, 0.25, 2 / 3]
    want, aux = _oracle_mid(sd, fr, ts)
    engine.prepare(fr[0].cuda().contiguous(), fr[1].cuda().contiguous())
    torch.cuda.synchronize()
    # intermediate taps first (sharper diagnostics than the final frame): PWC flows, refined flows, mask
    flow = engine.flow[0].cpu()
    for n, k in ((0, "fwd"), (1, "bwd")):
        wv = nhwc(aux[k])[0]
        assert (flow[n] - wv).abs().max().item() <= 1e-4, describe_diff(flow[n], wv, f"PWC {k} flow")
    d0, r = engine.d0.cpu(), engine.r.cpu()
    tf = nhwc(aux["ten_fwd"])[0]
    got_tf = torch.cat([d0[0, ..., 0:2] + r[0, ..., 2 * b:2 * b + 2] for b in range(4)], -1)
    assert (got_tf - tf).abs().max().item() <= 5e-4, describe_diff(got_tf, tf, "refined forward flows")
    wei = torch.sigmoid(r[0, ..., 8]) * 0.8 + 0.1
    assert (wei - aux["wei_f"][0, 0]).abs().max().item() <= 1e-4
    for t, wv in zip(ts, want):
        got = engine.render(t).cpu()
        assert got.shape == (h, w, 3)
        assert (got - wv).abs().max().item() <= TOL, describe_diff(got, wv, f"m2m t={t}")


def test_m2m_forward_1080p(engine, sd):
    """BASELINE.json's full size (pads to 1088x1920); the oracle needs ~6 s on the host for this."""
    fr = _frames(2, 1080, 1920, seed=9)
    want, _ = _oracle_mid(sd, fr, [0.5])
    got = engine.forward(fr[0].cuda().contiguous(), fr[1].cuda().contiguous(), 0.5).cpu()
    assert (got - want[0]).abs().max().item() <= TOL, describe_diff(got, want[0], "m2m 1080p")


def test_m2m_noise_and_alpha_channel(engine, sd):
    """i.i.d. noise input (worst-case gradients) with an RGBA tensor: channel 3 is ignored (vfi_utils.py:40-41)"""
    fr = synth.noise_frames(2, 70, 90, seed=4, c=4)
    want, _ = _oracle_mid(sd, fr, [0.5])
    got = engine.forward(fr[0].cuda().contiguous(), fr[1].cuda().contiguous(), 0.5).cpu()
    assert (got - want[0]).abs().max().item() <= TOL, describe_diff(got, want[0], "m2m noise")


def test_prepare_once_equals_per_timestep(engine):
    """prepare() is timestep independent: render(t) after other renders == a fresh forward at t (bit-exact up to the
    splat's atomic ordering, which the tolerance covers)"""
    fr = _frames(2, 96, 130, seed=5)
    a, b = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
    engine.prepare(a, b)
    x1 = engine.render(0.25).clone()
    _ = engine.render(0.75)
    x2 = engine.render(0.25).clone()
    x3 = engine.forward(a, b, 0.25)
    assert (x1 - x2).abs().max().item() <= 1e-5 and (x1 - x3).abs().max().item() <= 1e-5


@pytest.mark.parametrize("multiplier,states", [(2, None), (3, InterpolationStateList([1], True)), ([2, 0, 3], None),
                                               ([1, 2, 0], None), ([3], InterpolationStateList([0], False))])
def test_m2m_node_loop(engine, sd, multiplier, states):
    from cfi_amd.m2m import run_plan
    from cfi_amd.schedule import generic_output_plan

    fr = _frames(4, 64, 96, seed=7, shift=2.0)
    want = mo.m2m_vfi(sd, fr, multiplier, states)
    plan, tasks = generic_output_plan(len(fr), multiplier, states)
    got = run_plan(engine, fr, plan, tasks)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, "node")
    src = [i for i, (k, _) in enumerate(plan) if k == "src"]
    assert all(torch.equal(got[i], fr[plan[i][1]][..., :3]) for i in src), "pass-through frames must be bit-exact"


def test_m2m_vs_reference_golden(engine, golden_dir):
    """HIP path vs outputs of the reference's own M2M_PWC (tests/golden/m2m_net.npz, oracle/make_golden_film_m2m.py)"""
    import os

    import numpy as np

    g = np.load(os.path.join(golden_dir, "m2m_net.npz"))
    fr = torch.from_numpy(g["frames"])
    engine.prepare(fr[0].cuda().contiguous(), fr[1].cuda().contiguous())
    for k, t in enumerate(g["times"]):
        got, want = engine.render(float(t)).cpu(), torch.from_numpy(g["out"][k])
        assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"m2m vs reference golden t={t}")


@pytest.mark.parametrize("name,kw", [("m2", dict(multiplier=2)), ("m3_skip1", dict(multiplier=3, states=InterpolationStateList([1], True))),
                                     ("mlist_203", dict(multiplier=[2, 0, 3])), ("mlist_120", dict(multiplier=[1, 2, 0])),
                                     ("mlist_3_keep0", dict(multiplier=[3], states=InterpolationStateList([0], False)))])
def test_m2m_node_vs_reference_golden(sd, golden_dir, tmp_path, monkeypatch, name, kw):
    """The product node class end to end (checkpoint file -> M2M_VFI.vfi, RGBA input) vs the real reference node's output"""
    import os

    import numpy as np
    from cfi_amd import m2m

    g = np.load(os.path.join(golden_dir, "m2m_node.npz"))
    pth = tmp_path / "M2M.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(m2m, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    frames = torch.from_numpy(g["frames"])
    before = frames.clone()
    out = m2m.M2M_VFI().vfi("M2M.pth", frames, multiplier=kw["multiplier"], optional_interpolation_states=kw.get("states"))[0]
    want = torch.from_numpy(g[name])
    assert torch.equal(frames, before), "input must not be mutated"
    assert out.shape == want.shape and out.device.type == "cpu" and out.dtype == torch.float32
    assert (out - want).abs().max().item() <= TOL, describe_diff(out, want, "node vs reference golden")


def test_m2m_larger_flows(lib):
    """PWC flow heads scaled x4: refined flows of ~15 px at full resolution, a splat window of ~17 px around every tile."""
    from cfi_amd.m2m import M2MEngine

    sd = synth.m2m_synth_state_dict(99)
    big = {k: (v * 4.0 if ".netMain.netMain.10." in k else v) for k, v in sd.items()}
    eng = M2MEngine(big)
    try:
        fr = _frames(2, 160, 224, seed=8)
        want, aux = _oracle_mid(big, fr, [0.5, 0.3])
        fmax = aux["ten_fwd"].abs().max().item()
        assert fmax > 8.0, f"test premise: larger flows (got {fmax:.1f} px)"
        eng.prepare(fr[0].cuda().contiguous(), fr[1].cuda().contiguous())
        for t, wv in zip([0.5, 0.3], want):
            got = eng.render(t).cpu()
            assert (got - wv).abs().max().item() <= TOL, describe_diff(got, wv, f"m2m larger flows ({fmax:.0f} px) t={t}")
    finally:
        eng.close()
