"""-m gpu: the IFRNet node on the MI355X vs oracle/ifrnet_oracle.py (bit-exact vs the reference's
IRFNet_L / IRFNet_S and node here, oracle/VALIDATION_IFRNET.log) and vs outputs of the reference node
(tests/golden/ifrnet_node.npz).  Contract: per-pixel fp32 |d| <= 1e-3."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import describe_diff
from cfi_amd import synth
from cfi_amd.schedule import InterpolationStateList
from oracle import ifrnet_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _st():
    from cfi_amd import _lib

    return _lib.stream_ptr()


# ---- the kernels of csrc/ifrnet_ops.hip, one by one -----------------------------------------------------------------
def test_conv7x7s2_prelu(hip_lib):
    from cfi_amd import _lib

    torch.manual_seed(0)
    x = torch.randn(2, 3, 40, 56)
    w, b, sl = torch.randn(64, 3, 7, 7) * 0.1, torch.randn(64) * 0.1, torch.rand(64) * 0.3 + 0.1
    want = F.prelu(F.conv2d(x, w, b, 2, 3), sl).permute(0, 2, 3, 1).contiguous()
    xin = torch.zeros(2, 40, 56, 8)
    xin[..., :3] = x.permute(0, 2, 3, 1)
    xin, wd, bd, sd_ = xin.cuda(), w.permute(2, 3, 1, 0).contiguous().cuda(), b.cuda(), sl.cuda()
    out = torch.zeros(2, 20, 28, 64, device="cuda")
    _lib.check(hip_lib.vfi_conv7x7s2_prelu(xin.data_ptr(), 8, wd.data_ptr(), bd.data_ptr(), sd_.data_ptr(), 64, out.data_ptr(), 64, 2, 40,
                                           56, _st()), "conv7x7")
    assert (out.cpu() - want).abs().max().item() <= 2e-5, describe_diff(out.cpu(), want, "conv7x7s2")


@pytest.mark.parametrize("hin,win,s", [(64, 128, 0.5), (128, 64, 0.75), (96, 48, 1.0 / 0.75), (32, 32, 4.0), (48, 80, 1.0)])
def test_resize_bilinear_ratio(hip_lib, hin, win, s):
    """F.interpolate(scale_factor=s): output size floor(in*s), source step (float)(1/s)"""
    from cfi_amd import _lib

    torch.manual_seed(1)
    x = torch.randn(2, 5, hin, win)
    want = (F.interpolate(x, scale_factor=s, mode="bilinear", align_corners=False) * 1.5).permute(0, 2, 3, 1).contiguous()
    ho, wo = want.shape[1:3]
    xin = torch.zeros(2, hin, win, 8)
    xin[..., 2:7] = x.permute(0, 2, 3, 1)
    xin = xin.cuda()
    out = torch.zeros(2, ho, wo, 8, device="cuda")
    _lib.check(hip_lib.vfi_resize_bilinear_ratio(xin.data_ptr() + 8, 8, out.data_ptr() + 4, 8, 2, hin, win, ho, wo, 5, 1.0 / s, 1.0 / s, 1.5,
                                                 _st()), "resize_ratio")
    got = out.cpu()
    assert (got[..., 1:6] - want).abs().max().item() <= 1e-5, describe_diff(got[..., 1:6], want, f"resize x{s}")
    assert got[..., 0].abs().max() == 0 and got[..., 6:].abs().max() == 0


def test_prep_center_fill_sigmoid(hip_lib):
    from cfi_amd import _lib

    N, H, W, Hp, Wp = 2, 50, 70, 64, 128
    fr = synth.noise_frames(2 * N, H, W, seed=3, c=4).cuda()
    img = torch.full((2 * N, Hp, Wp, 4), 7.0, device="cuda")
    for n in range(N):
        _lib.check(hip_lib.vfi_ifrnet_prep(fr[2 * n].data_ptr(), fr[2 * n + 1].data_ptr(), 4, H, W, img[n].data_ptr(), img[N + n].data_ptr(),
                                           Hp, Wp, _st()), "prep")
    i0 = F.pad(fr[0::2, ..., :3].cpu().permute(0, 3, 1, 2), (0, Wp - W, 0, Hp - H))
    i1 = F.pad(fr[1::2, ..., :3].cpu().permute(0, 3, 1, 2), (0, Wp - W, 0, Hp - H))
    got = img.cpu()
    assert torch.equal(got[:N, ..., :3], i0.permute(0, 2, 3, 1)) and torch.equal(got[N:, ..., :3], i1.permute(0, 2, 3, 1))
    assert got[..., 3].abs().max() == 0
    rm, cm, mean = torch.zeros(2 * N, Hp, 4, device="cuda"), torch.zeros(2 * N, 4, device="cuda"), torch.zeros(N, device="cuda")
    _lib.check(hip_lib.vfi_pool_mean(img.data_ptr(), 4, rm.data_ptr(), 4, 2 * N, Hp, Wp, 4, 1, _st()), "pool")
    _lib.check(hip_lib.vfi_pool_mean(rm.data_ptr(), 4, cm.data_ptr(), 4, 2 * N, Hp, 1, 4, 0, _st()), "pool")
    _lib.check(hip_lib.vfi_ifrnet_center(img.data_ptr(), cm.data_ptr(), mean.data_ptr(), N, Hp * Wp, _st()), "center")
    m = torch.cat([i0, i1], 2).mean(1, keepdim=True).mean(2, keepdim=True).mean(3, keepdim=True)
    assert (mean.cpu() - m.view(-1)).abs().max().item() <= 1e-6
    got = img.cpu()
    assert (got[:N, ..., :3] - (i0 - m).permute(0, 2, 3, 1)).abs().max().item() <= 1e-6
    assert (got[N:, ..., :3] - (i1 - m).permute(0, 2, 3, 1)).abs().max().item() <= 1e-6
    # fill + sigmoid on channel windows
    t = torch.randn(3, 6, 10, 8, device="cuda")
    ref = t.cpu().clone()
    vals = (C.c_float * 3)(0.25, 0.5, 1.0)
    _lib.check(hip_lib.vfi_fill_items(t.data_ptr() + 4 * 5, 8, 2, 3, 60, vals, _st()), "fill")
    _lib.check(hip_lib.vfi_sigmoid(t.data_ptr() + 4 * 1, 8, 3, 3 * 60, _st()), "sigmoid")
    for n, v in enumerate((0.25, 0.5, 1.0)):
        ref[n, ..., 5:7] = v
    ref[..., 1:4] = torch.sigmoid(ref[..., 1:4])
    assert (t.cpu() - ref).abs().max().item() <= 1e-6


@pytest.mark.parametrize("hf,wf", [(64, 128), (63, 127)])
def test_ifrnet_output(hip_lib, hf, wf):
    """both image warps with a flow field that may be smaller than the padded images + merge + crop"""
    from cfi_amd import _lib

    torch.manual_seed(2)
    N, Hp, Wp, H, W = 2, 64, 128, 50, 100
    img = torch.randn(2 * N, 3, Hp, Wp) * 0.3
    flow = torch.randn(N, 4, hf, wf) * 4.0
    mask, res, mean = torch.rand(N, 1, hf, wf), torch.randn(N, 3, hf, wf) * 0.1, torch.rand(N, 1, 1, 1)
    want = torch.clamp(mask * ifrnet_oracle.warp(img[:N], flow[:, 0:2]) + (1 - mask) * ifrnet_oracle.warp(img[N:], flow[:, 2:4]) + mean + res,
                       0, 1)[:, :, :H, :W].permute(0, 2, 3, 1).contiguous()
    im = torch.zeros(2 * N, Hp, Wp, 4)
    im[..., :3] = img.permute(0, 2, 3, 1)
    fin = torch.cat([flow, mask, res], 1).permute(0, 2, 3, 1).contiguous()
    im, fin, md = im.cuda(), fin.cuda(), mean.view(-1).contiguous().cuda()
    out = torch.zeros(N, H, W, 3, device="cuda")
    _lib.check(hip_lib.vfi_ifrnet_output(im[:N].data_ptr(), im[N:].data_ptr(), fin.data_ptr(), md.data_ptr(), out.data_ptr(), N, Hp, Wp, hf,
                                         wf, H, W, _st()), "output")
    assert (out.cpu() - want).abs().max().item() <= 1e-5, describe_diff(out.cpu(), want, "ifrnet_output")


# ---- the network ----------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=["L", "S"])
def net(request, hip_lib):
    from cfi_amd.ifrnet import IFRNetEngine

    torch.cuda.set_device(0)
    kind = request.param
    sd = synth.ifrnet_synth_state_dict(kind, 1234)
    e = IFRNetEngine(sd, kind)
    yield kind, sd, e
    e.close()


@pytest.mark.parametrize("h,w,n,sf,t", [(64, 64, 1, 1.0, 0.5), (100, 150, 2, 0.5, 1.0), (200, 328, 1, 0.5, 1.0), (72, 100, 1, 0.75, 1.0),
                                        (128, 192, 1, 0.25, 0.3), (64, 64, 1, 0.5, 1.0)])
def test_engine_against_oracle(net, h, w, n, sf, t):
    kind, sd, e = net
    fr = synth.smooth_frames(2 * n, h, w, seed=h + n, shift=2.5)
    i0, i1 = fr[0::2].permute(0, 3, 1, 2).contiguous(), fr[1::2].permute(0, 3, 1, 2).contiguous()
    with torch.inference_mode():
        want = ifrnet_oracle.ifrnet_forward(sd, i0, i1, sf, t).permute(0, 2, 3, 1).contiguous()
    dev = fr.cuda()
    out = torch.empty(n, h, w, 3, device="cuda")
    e.forward([dev[2 * k] for k in range(n)], [dev[2 * k + 1] for k in range(n)], sf, t, out)
    got = out.cpu()
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"IFRNet_{kind} {h}x{w} sf={sf} t={t}")


def test_engine_1080p_node_default(net):
    """The node's default call at 1080p (multiplier 2 -> working resolution 0.5, time embedding = scale_factor widget 1.0)
    against the oracle at full size (measured: 5e-5 max, profiles/r01e_ifrnet_bench_1080p.txt)."""
    kind, sd, e = net
    fr = synth.smooth_frames(2, 1080, 1920, seed=2, shift=4.0)
    x = fr.permute(0, 3, 1, 2)
    with torch.inference_mode():
        want = ifrnet_oracle.ifrnet_forward(sd, x[0:1], x[1:2], 0.5, 1.0).permute(0, 2, 3, 1).contiguous()
    dev = fr.cuda()
    out = torch.empty(1, 1080, 1920, 3, device="cuda")
    e.forward([dev[0]], [dev[1]], 0.5, 1.0, out)
    got = out.cpu()
    e.release_workspace()
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"IFRNet_{kind} 1080p")


def test_engine_rejects_what_the_reference_rejects(net):
    kind, sd, e = net
    fr = synth.smooth_frames(2, 64, 64, seed=1).cuda()
    out = torch.empty(1, 64, 64, 3, device="cuda")
    with pytest.raises(RuntimeError, match="multiple of 16"):
        e.forward([fr[0]], [fr[1]], 1.0 / 3.0, 1.0, out)        # 21x21 working resolution: torch.cat fails in the reference


NODE_CASES = {
    "x2": dict(multiplier=2),
    "x2_t05": dict(multiplier=2, scale_factor=0.5),
    "x4_skip1": dict(multiplier=4, optional_interpolation_states=InterpolationStateList([1], True)),
}


@pytest.mark.parametrize("kind", ["L", "S"])
@pytest.mark.parametrize("name", list(NODE_CASES))
def test_node_against_reference_golden(hip_lib, golden_dir, tmp_path, monkeypatch, kind, name):
    """IFRNet_VFI.vfi — same call as the reference's node — vs outputs of the reference node"""
    import cfi_amd.ifrnet as I
    from cfi_amd import ckpt

    pth = tmp_path / f"IFRNet_{kind}_Vimeo90K.pth"
    torch.save(synth.ifrnet_synth_state_dict(kind, 1234), pth)
    monkeypatch.setattr(I, "load_file_from_github_release", lambda model_type, ckpt_name: str(pth))
    g = np.load(os.path.join(golden_dir, "ifrnet_node.npz"))
    frames = torch.from_numpy(g[f"{kind}_frames"])
    before = frames.clone()
    try:
        (out,) = I.IFRNet_VFI().vfi(pth.name, frames, clear_cache_after_n_frames=10, **NODE_CASES[name])
    finally:
        ckpt.clear_engine_cache()
    want = torch.from_numpy(g[f"{kind}_{name}"])
    assert torch.equal(frames, before), "input tensor was mutated"
    assert out.dtype == torch.float32 and out.device.type == "cpu" and out.shape == want.shape
    assert (out - want).abs().max().item() <= TOL, describe_diff(out, want, f"IFRNet_{kind} node {name}")


@pytest.mark.parametrize("kind", ["L", "S"])
def test_node_long_clip_keeps_frames_until_the_last_render(hip_lib, tmp_path, monkeypatch, kind):
    """9 frames x4: more frames than the upload ring has slots (4) and three renders per pair.  IFRNetEngine.prepare() only
    keeps references to the ring-slot tensors, so a slot released before the pair's last render would be overwritten by a
    later frame.  Every new frame must equal the engine run on that pair alone."""
    import cfi_amd.ifrnet as I
    from cfi_amd import ckpt
    from cfi_amd.ifrnet import IFRNetEngine

    sd = synth.ifrnet_synth_state_dict(kind, 1234)
    pth = tmp_path / f"IFRNet_{kind}_Vimeo90K.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(I, "load_file_from_github_release", lambda model_type, ckpt_name: str(pth))
    frames = torch.cat([synth.smooth_frames(3, 128, 192, seed=s, shift=3.0) for s in (1, 2, 3)])     # 9 distinct frames
    try:
        (out,) = I.IFRNet_VFI().vfi(pth.name, frames, clear_cache_after_n_frames=10, multiplier=4)
    finally:
        ckpt.clear_engine_cache()
    assert out.shape == (33, 128, 192, 3)
    e = IFRNetEngine(sd, kind)
    tmp = torch.empty(1, 128, 192, 3, device="cuda")
    for pair in range(8):
        a, b = frames[pair].cuda().contiguous(), frames[pair + 1].cuda().contiguous()
        for k in (1, 2, 3):
            if (128 * k) % 64 or (192 * k) % 64:      # working resolution k/4 must be a multiple of 16
                continue
            e.forward([a], [b], k / 4, 1.0, tmp)
            want, got = tmp[0].cpu(), out[4 * pair + k]
            assert (got - want).abs().max().item() <= 1e-6, describe_diff(got, want, f"IFRNet_{kind} pair {pair} k {k}")
        assert torch.equal(out[4 * pair], frames[pair])
    e.close()
