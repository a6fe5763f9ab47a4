"""CPU: the orchestration of comfyui-frame-interpolation_amd/ifunet.py (IFUNetEngine.forward) against oracle/ifunet_oracle.py
through the test double of the C ABI (tests/emu_backend.py: the real IFUNet / GMFSS kernel bodies on the host + torch
restatements of the older entry points)."""
import pytest
import torch

from cfi_amd import synth
from emu_backend import EmuBackend
from oracle import ifunet_oracle as O


@pytest.fixture(scope="module")
def setup():
    from cfi_amd.ifunet import IFUNetEngine

    sd = synth.ifunet_synth_state_dict(1234)
    eng = IFUNetEngine(sd, _test_backend=EmuBackend())
    yield sd, eng
    eng.close()


def check_against_oracle(eng, sd, fr, t, scale, ensemble, out):
    """shared with tests/test_gpu_ifunet.py"""
    x = fr.permute(0, 3, 1, 2).contiguous()
    with torch.inference_mode():
        want = O.ifunet_forward(sd, x[0:1], x[1:2], t, scale, ensemble).permute(0, 2, 3, 1)[0]
    dev = eng.device
    eng.forward(fr[0].contiguous().to(dev), fr[1].contiguous().to(dev), t, out, scale=scale, ensemble=ensemble)
    d = (out.cpu() - want).abs()
    return d.max().item(), d.mean().item()


@pytest.mark.parametrize("h,w,t,scale,ens", [(64, 64, 0.5, 1.0, False), (100, 150, 0.25, 1.0, True), (128, 128, 0.5, 0.5, True),
                                             (64, 64, 0.75, 2.0, False)])
def test_forward_matches_oracle(setup, h, w, t, scale, ens):
    sd, eng = setup
    fr = synth.smooth_frames(2, h, w, seed=h + 3, shift=2.5)
    mx, mean = check_against_oracle(eng, sd, fr, t, scale, ens, torch.zeros(h, w, 3))
    assert mx <= 1e-3, f"IFUNet {h}x{w} t={t} scale={scale} ensemble={ens}: max {mx} mean {mean}"
    eng.release_workspace()


def test_rejects_scales_the_device_path_does_not_cover(setup):
    sd, eng = setup
    fr = synth.smooth_frames(2, 64, 64, seed=1)
    with pytest.raises(NotImplementedError):
        eng.forward(fr[0].contiguous(), fr[1].contiguous(), 0.5, torch.zeros(64, 64, 3), scale=0.3)


NODE_CASES = {
    "x2": dict(multiplier=2),
    "x2_noens_s05": dict(multiplier=2, scale_factor=0.5, ensemble=False),
}


@pytest.mark.parametrize("name", list(NODE_CASES))
def test_node_against_reference_golden_on_the_test_double(golden_dir, tmp_path, monkeypatch, name):
    """IFUnet_VFI.vfi — same call as the reference's node, engine on the CPU test double — vs outputs of the reference node"""
    import os

    import numpy as np

    import cfi_amd.ckpt as K
    import cfi_amd.ifunet as M

    pth = tmp_path / "IFUNet.pth"
    torch.save(synth.ifunet_synth_state_dict(1234), pth)
    monkeypatch.setattr(K, "load_file_from_github_release", lambda model_type, ckpt_name: str(pth))
    real = M.IFUNetEngine
    monkeypatch.setattr(M, "IFUNetEngine", lambda sd: real(sd, _test_backend=EmuBackend()))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    g = np.load(os.path.join(golden_dir, "ifunet_node.npz"))
    frames = torch.from_numpy(g["frames"])
    (out,) = M.IFUnet_VFI().vfi("IFUNet.pth", frames, clear_cache_after_n_frames=10, **NODE_CASES[name])
    want = torch.from_numpy(g[name])
    assert out.shape == want.shape and out.dtype == torch.float32
    assert (out - want).abs().max().item() <= 1e-3
    assert torch.equal(out[0], frames[0]) and torch.equal(out[-1], frames[-1])


def test_repeated_calls_are_bit_identical_with_recycled_scratch():
    """the engine's scratch comes from a pool and is recycled stage by stage: later calls run on stale, non-zero blocks — same
    output bit for bit, and the pool stops growing after the first call of a shape"""
    from cfi_amd.ifunet import IFUNetEngine

    eng = IFUNetEngine(synth.ifunet_synth_state_dict(77), _test_backend=EmuBackend())
    try:
        fr = synth.smooth_frames(3, 64, 96, seed=5, shift=2.0)
        outs = []
        for rep, (a, b, t, ens) in enumerate(((0, 1, 0.5, True), (1, 2, 0.25, False), (0, 1, 0.5, True))):
            o = torch.zeros(64, 96, 3)
            eng.forward(fr[a].contiguous(), fr[b].contiguous(), t, o, scale=1.0, ensemble=ens)
            outs.append(o)
            if rep == 0:
                used = eng.workspace_bytes()
        assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1])
        assert eng.workspace_bytes() == used, "the pool keeps growing"
        assert eng._live == [{}], "scratch outlived the call"
    finally:
        eng.close()
