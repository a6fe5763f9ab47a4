"""-m gpu: FILM path on the MI355X vs the oracle (oracle/film_oracle.py, bit-exact vs the reference's film_arch
Interpolator in the build container).  Generic ops first, then the whole interpolator, then the node."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import describe_diff, nhwc, ptr
from cfi_amd import synth
from oracle import film_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    from cfi_amd import _lib

    _lib.check(hip_lib.vfi_init(0), "vfi_init")
    return hip_lib


@pytest.fixture(scope="module")
def sd():
    return synth.film_synth_state_dict(1234)


@pytest.fixture(scope="module")
def engine(lib, sd):
    from cfi_amd.film import FilmEngine

    e = FilmEngine(sd)
    yield e
    e.close()


def _ck(rc, what):
    from cfi_amd import _lib

    _lib.check(rc, what)


@pytest.mark.parametrize("k,cin,cout,h,w", [(3, 3, 64, 37, 45), (3, 202, 64, 30, 41), (2, 128, 64, 33, 40), (2, 1930, 512, 9, 15),
                                            (1, 256, 128, 20, 31), (1, 64, 3, 35, 50), (1, 16, 2, 17, 23), (3, 1920, 256, 8, 15),
                                            (3, 32, 32, 40, 60), (3, 896, 128, 17, 30)])
def test_generic_conv_same(lib, k, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = torch.rand(1, cin, h, w, generator=g) * 2 - 1
    wt = (torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5
    b = torch.rand(cout, generator=g) - 0.5
    want = nhwc(F.leaky_relu(F.conv2d(x, wt, b, padding="same"), 0.2))
    cphys = (cin + 7) // 8 * 8
    # physical input: channels in REVERSE order plus zero padding, to exercise chan_map; written inside a wider tensor
    cmap = [cphys - 1 - c for c in range(cin)]
    xin = torch.zeros(h, w, cphys + 8)
    xin[..., 4:4 + cphys][..., cmap] = x[0].permute(1, 2, 0)
    xd = xin.cuda()
    out = torch.full((h, w, cout + 5), float("nan"), device="cuda")
    import ctypes as C
    cm = (C.c_int * cin)(*cmap)
    hnd = lib.vfi_conv_create(wt.data_ptr(), b.data_ptr(), cout, cin, k, k, cm, cphys)
    assert hnd, "create failed"
    _ck(lib.vfi_conv_forward(hnd, xd.data_ptr() + 16, cphys + 8, out.data_ptr() + 12, cout + 5, 1, h, w, 1, 0.2, None), "conv")
    torch.cuda.synchronize()
    lib.vfi_conv_destroy(hnd)
    got = out.cpu()
    assert torch.isnan(got[..., :3]).all() and torch.isnan(got[..., 3 + cout:]).all()
    got = got[..., 3:3 + cout][None]
    tol = 2e-5 * max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"conv{k}x{k} {cin}->{cout}")


@pytest.mark.parametrize("cin,cout,h,w,act,cmapped", [(128, 64, 33, 40, 0, False), (256, 128, 17, 23, 1, False), (202, 64, 9, 15, 0, True), (64, 64, 64, 96, 0, False),
                                                    (512, 256, 20, 31, 0, False)])
def test_up2x2_layer_equals_nearest_upsample_then_conv2x2(lib, cin, cout, h, w, act, cmapped):
    """vfi_conv_create_up2x2 (r6): F.interpolate(x, scale_factor=2, mode='nearest') + Conv2d(2, padding='same') [+ LeakyReLU 0.2] of FILM's Fusion
    (film_arch.py:282-292) computed on the low-resolution tensor — against torch on the host, and against this library's own two-step form."""
    import ctypes as C

    g = torch.Generator().manual_seed(cin * 3 + cout + h)
    x = torch.rand(1, cin, h, w, generator=g) * 2 - 1
    wt = (torch.rand(cout, cin, 2, 2, generator=g) * 2 - 1) / (cin * 4) ** 0.5
    b = torch.rand(cout, generator=g) - 0.5
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding="same")
    want = nhwc(F.leaky_relu(y, 0.2) if act else y)
    cphys = (cin + 7) // 8 * 8
    cmap = [cphys - 1 - c for c in range(cin)] if cmapped else list(range(cin))
    xin = torch.zeros(h, w, cphys + 8)
    xin[..., 4:4 + cphys][..., cmap] = x[0].permute(1, 2, 0)
    xd = xin.cuda()
    cm = (C.c_int * cin)(*cmap)
    hnd = lib.vfi_conv_create_up2x2(wt.data_ptr(), b.data_ptr(), cout, cin, cm, cphys)
    assert hnd, "create failed"
    out = torch.full((2 * h, 2 * w, cout + 8), float("nan"), device="cuda")
    _ck(lib.vfi_conv_forward(hnd, xd.data_ptr() + 16, cphys + 8, out.data_ptr() + 16, cout + 8, 1, h, w, act, 0.2, None), "up2x2 forward")
    torch.cuda.synchronize()
    lib.vfi_conv_destroy(hnd)
    got = out.cpu()
    assert torch.isnan(got[..., :4]).all() and torch.isnan(got[..., 4 + cout:]).all(), "wrote outside its channel window"
    got = got[..., 4:4 + cout][None]
    tol = 2e-5 * max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"up2x2 {cin}->{cout} @{h}x{w}")
    # the two-step form through the same library (what FILM ran until round 6)
    h2 = lib.vfi_conv_create(wt.data_ptr(), b.data_ptr(), cout, cin, 2, 2, cm, cphys)
    up = torch.zeros((2 * h, 2 * w, cphys), device="cuda")
    _ck(lib.vfi_upsample_nearest(xd.data_ptr() + 16, cphys + 8, up.data_ptr(), cphys, 1, h, w, 2 * h, 2 * w, cphys, None), "upsample")
    out2 = torch.empty((2 * h, 2 * w, cout), device="cuda")
    _ck(lib.vfi_conv_forward(h2, up.data_ptr(), cphys, out2.data_ptr(), cout, 1, 2 * h, 2 * w, act, 0.2, None), "conv2x2")
    torch.cuda.synchronize()
    lib.vfi_conv_destroy(h2)
    assert (got[0] - out2.cpu()).abs().max().item() <= tol, "up2x2 vs upsample + conv2x2 of this library"


def test_avgpool_nearest_bilinear_axpby(lib):
    g = torch.Generator().manual_seed(3)
    x = torch.rand(1, 8, 33, 45, generator=g)
    xd = nhwc(x).cuda()
    o = torch.empty(1, 16, 22, 8, device="cuda")
    _ck(lib.vfi_avgpool2(ptr(xd), 8, ptr(o), 8, 1, 33, 45, 8, None), "avgpool")
    assert torch.equal(o.cpu(), nhwc(F.avg_pool2d(x, 2, 2)))
    o = torch.empty(1, 67, 91, 8, device="cuda")
    _ck(lib.vfi_upsample_nearest(ptr(xd), 8, ptr(o), 8, 1, 33, 45, 67, 91, 8, None), "nearest")
    assert torch.equal(o.cpu(), nhwc(F.interpolate(x, size=(67, 91), mode="nearest")))
    v = torch.rand(1, 2, 33, 60, generator=g) * 10 - 5
    vd = nhwc(v).cuda()
    o = torch.empty(1, 67, 120, 2, device="cuda")
    _ck(lib.vfi_resize_bilinear(ptr(vd), 2, ptr(o), 2, 1, 33, 60, 67, 120, 2, 2.0, None), "bilinear")
    want = nhwc(F.interpolate(2 * v, size=(67, 120), mode="bilinear"))
    assert (o.cpu() - want).abs().max().item() <= 2e-6, describe_diff(o.cpu(), want, "bilinear")
    a, b = torch.rand(5, 7, 6), torch.rand(5, 7, 4)
    o = torch.zeros(5, 7, 9, device="cuda")
    ad, bd = a.cuda(), b.cuda()   # keep the device tensors alive across the asynchronous launch
    _ck(lib.vfi_axpby(ad.data_ptr() + 4, 6, bd.data_ptr() + 8, 4, o.data_ptr() + 12, 9, 35, 2, 0.5, 2.0, None), "axpby")
    torch.cuda.synchronize()
    want = torch.zeros(5, 7, 9)
    want[..., 3:5] = 0.5 * a[..., 1:3] + 2.0 * b[..., 2:4]
    assert torch.equal(o.cpu(), want)


@pytest.mark.parametrize("shape,mag", [((1, 8, 40, 56), 30.0), ((1, 68, 67, 120), 12.0), ((1, 4, 135, 240), 300.0)])
def test_warp_film_vs_oracle(lib, shape, mag):
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h)
    x = torch.rand(n, c, h, w, generator=g)
    fl = (torch.rand(n, 2, h, w, generator=g) - 0.5) * mag
    want = nhwc(film_oracle.warp(x, fl * 0.5))
    xd, fd = nhwc(x).cuda(), nhwc(fl).cuda()
    out = torch.empty_like(xd)
    _ck(lib.vfi_warp_film(ptr(xd), c, ptr(fd), 2, 0.5, ptr(out), c, n, h, w, c, None), "warp_film")
    got = out.cpu()
    assert (got - want).abs().max().item() <= 3e-6, describe_diff(got, want, "warp_film")


@pytest.mark.parametrize("h,w", [(64, 96), (135, 240), (270, 480)])
def test_interpolator_vs_oracle(engine, sd, h, w):
    fr = synth.smooth_frames(2, h, w, seed=h, shift=2.0)
    got = engine.forward(fr[0].cuda().contiguous(), fr[1].cuda().contiguous()).cpu()
    x0 = fr[0:1].permute(0, 3, 1, 2).contiguous()
    x1 = fr[1:2].permute(0, 3, 1, 2).contiguous()
    with torch.inference_mode():
        want, aux = film_oracle.film_forward(sd, x0, x1, return_aux=True)
    want = want[0].permute(1, 2, 0)
    msgs = []
    for d, name in ((0, "fwd_flow"), (1, "bwd_flow")):
        for l in (4, 2, 0):
            a = engine.debug_flow(d, l, h >> l, w >> l)
            b = aux[name][l][0].permute(1, 2, 0)
            msgs.append(describe_diff(a, b, f"{name}[{l}]"))
    msgs.append(describe_diff(got, want, "output"))
    assert (got - want).abs().max().item() <= 1e-3, "\n".join(msgs)


def test_film_node_schedule_and_skip(lib, sd, tmp_path, monkeypatch):
    import cfi_amd.film as FM
    from cfi_amd.schedule import InterpolationStateList

    pth = tmp_path / "film_net_fp32.pt"
    torch.save(sd, pth)
    monkeypatch.setattr(FM, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    frames = synth.smooth_frames(3, 64, 80, seed=2, shift=1.5, c=4)
    (out,) = FM.FILM_VFI().vfi("film_net_fp32.pt", frames, multiplier=3)
    want = film_oracle.film_vfi(sd, frames, multiplier=3)
    assert out.shape == want.shape == (7, 64, 80, 3)
    assert (out - want).abs().max().item() <= 1e-3, describe_diff(out, want, "film node x3")
    assert torch.equal(out[0], frames[0, ..., :3]) and torch.equal(out[-1], frames[-1, ..., :3])
    (out,) = FM.FILM_VFI().vfi("film_net_fp32.pt", frames, multiplier=2,
                               optional_interpolation_states=InterpolationStateList([0], True))
    want = film_oracle.film_vfi(sd, frames, multiplier=2, states=InterpolationStateList([0], True))
    assert out.shape == want.shape == (3, 64, 80, 3)   # pair 0 dropped entirely (reference quirk)
    assert (out - want).abs().max().item() <= 1e-3


def test_interpolator_vs_reference_golden(engine, golden_dir):
    """HIP path vs the output of the reference's own film_arch.Interpolator (tests/golden/film_net.npz, written by
    oracle/make_golden_film_m2m.py in the build container)"""
    g = np.load(os.path.join(golden_dir, "film_net.npz"))
    fr = torch.from_numpy(g["frames"])
    got = engine.forward(fr[0].cuda().contiguous(), fr[1].cuda().contiguous()).cpu()
    want = torch.from_numpy(g["out"])[0]
    assert (got - want).abs().max().item() <= 1e-3, describe_diff(got, want, "film vs reference golden")


def test_config2_film_1080p(engine, sd):
    """BASELINE.json configs[2]: FILM 2x at 1080x1920 (one Interpolator call), against the oracle (~35 s on the host)."""
    fr = synth.smooth_frames(2, 1080, 1920, seed=2, shift=4.0)
    x = fr.permute(0, 3, 1, 2).contiguous()
    with torch.inference_mode():
        want = film_oracle.film_forward(sd, x[0:1], x[1:2])[0].permute(1, 2, 0)
    got = engine.forward(fr[0].cuda().contiguous(), fr[1].cuda().contiguous()).cpu()
    assert (got - want).abs().max().item() <= 1e-3, describe_diff(got, want, "film 1080p")


@pytest.mark.parametrize("h,w", [(64, 80), (270, 480), (1080, 1920)])
def test_two_stream_forward_equals_one_stream(lib, sd, h, w):
    """r6: vfi_film_forward runs image 1's feature extraction and the backward flow pyramid on the object's side stream (option film_side)
    beside image 0's and the forward one.  Same kernels on the same tensors: bit-identical frames, call after call (the next call's side
    work must wait for the previous call's fusion)."""
    from cfi_amd.film import FilmEngine

    fr = synth.smooth_frames(3, h, w, seed=4, shift=2.0)
    x = [f.cuda().contiguous() for f in fr]
    eng = FilmEngine(sd)
    try:
        outs = {}
        for mode in (1, 0, 1, 1):
            assert lib.vfi_test_set_option(b"film_side", mode) == 0
            outs.setdefault(mode, []).append([eng.forward(x[a], x[b]).cpu() for a, b in ((0, 1), (1, 2), (2, 0), (0, 1))])
        for k in range(4):
            assert torch.equal(outs[1][0][k], outs[0][0][k]), describe_diff(outs[1][0][k], outs[0][0][k], f"two streams vs one, call {k}")
            assert torch.equal(outs[1][0][k], outs[1][1][k]) and torch.equal(outs[1][0][k], outs[1][2][k]), "run-to-run determinism"
        assert torch.equal(outs[1][0][0], outs[1][0][3]), "the same pair again"
        assert eng.two_streams(False) is True      # the product switch (vfi_film_two_streams): the node turns the fork off under pair lanes
        one = eng.forward(x[0], x[1]).cpu()
        assert eng.two_streams(True) is False and torch.equal(one, outs[1][0][0])
    finally:
        lib.vfi_test_set_option(b"film_side", 1)
        eng.close()
