"""-m gpu: IFUNet (second half) on the MI355X against oracle/ifunet_oracle.py and the reference node's
goldens (ifunet/__init__.py:32-58, IFUNet_arch.py:364-503,627-638,764).  First ran on an MI355X in round 2
(profiles/r02_ifunet_first_gpu_run.txt): 1.4e-5 max vs the oracle at 256x448."""
import os

import numpy as np
import pytest
import torch

from cfi_amd import synth
from gpu_util import describe_diff
from test_ifunet_engine_cpu import NODE_CASES, check_against_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(hip_lib):
    from cfi_amd.ifunet import IFUNetEngine

    torch.cuda.set_device(0)
    sd = synth.ifunet_synth_state_dict(1234)
    eng = IFUNetEngine(sd)
    yield sd, eng
    eng.close()


@pytest.mark.parametrize("h,w,t,scale,ens", [(64, 64, 0.5, 1.0, False), (100, 150, 0.25, 1.0, True), (128, 128, 0.5, 0.5, True),
                                             (200, 328, 0.5, 1.0, True), (64, 64, 0.75, 2.0, False)])
def test_forward_matches_oracle(setup, h, w, t, scale, ens):
    sd, eng = setup
    fr = synth.smooth_frames(2, h, w, seed=h + 3, shift=2.5)
    mx, mean = check_against_oracle(eng, sd, fr, t, scale, ens, torch.zeros(h, w, 3, device="cuda"))
    assert mx <= 1e-3, f"IFUNet {h}x{w} t={t} scale={scale} ensemble={ens}: max {mx} mean {mean}"
    eng.release_workspace()


@pytest.mark.parametrize("ens", [True, False])
def test_forward_1080p(setup, ens):
    """The node's default call (and ensemble off) at 1080x1920 against the oracle at full size."""
    sd, eng = setup
    fr = synth.smooth_frames(2, 1080, 1920, seed=2, shift=4.0)
    mx, mean = check_against_oracle(eng, sd, fr, 0.5, 1.0, ens, torch.zeros(1080, 1920, 3, device="cuda"))
    eng.release_workspace()
    assert mx <= 1e-3, f"IFUNet 1080p ensemble={ens}: max {mx} mean {mean}"


@pytest.mark.parametrize("name", list(NODE_CASES))
def test_node_against_reference_golden(hip_lib, golden_dir, tmp_path, monkeypatch, name):
    """IFUnet_VFI.vfi — same call as the reference's node — vs outputs of the reference node (oracle/make_golden*.py)"""
    import cfi_amd.ckpt as K
    import cfi_amd.ifunet as M

    pth = tmp_path / "IFUNet.pth"
    torch.save(synth.ifunet_synth_state_dict(1234), pth)
    monkeypatch.setattr(K, "load_file_from_github_release", lambda model_type, ckpt_name: str(pth))
    g = np.load(os.path.join(golden_dir, "ifunet_node.npz"))
    frames = torch.from_numpy(g["frames"])
    before = frames.clone()
    (out,) = M.IFUnet_VFI().vfi("IFUNet.pth", frames, clear_cache_after_n_frames=10, **NODE_CASES[name])
    want = torch.from_numpy(g[name])
    assert torch.equal(frames, before), "input tensor was mutated"
    assert out.shape == want.shape and out.dtype == torch.float32 and out.device.type == "cpu"
    assert (out - want).abs().max().item() <= 1e-3, describe_diff(out, want, f"IFUnet node {name}")
    assert torch.equal(out[0], frames[0]) and torch.equal(out[-1], frames[-1])


def test_node_long_clip_keeps_frames_until_the_last_render(hip_lib, setup, tmp_path, monkeypatch):
    """9 frames x3: more frames than the upload ring has slots (4), several renders per pair.  prepare() only keeps
    references to the ring-slot tensors, so a slot released before the pair's last render would be overwritten by a later
    frame.  Every new frame must equal the engine run on that pair alone."""
    import cfi_amd.ckpt as K
    import cfi_amd.ifunet as M

    sd, eng = setup
    pth = tmp_path / "IFUNet.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(K, "load_file_from_github_release", lambda model_type, ckpt_name: str(pth))
    frames = torch.cat([synth.smooth_frames(3, 96, 128, seed=s, shift=3.0) for s in (1, 2, 3)])     # 9 distinct frames
    (out,) = M.IFUnet_VFI().vfi("IFUNet.pth", frames, clear_cache_after_n_frames=10, multiplier=3, ensemble=False)
    assert out.shape == (25, 96, 128, 3)
    tmp = torch.empty(96, 128, 3, device="cuda")
    for pair in range(8):
        a, b = frames[pair].cuda().contiguous(), frames[pair + 1].cuda().contiguous()
        for k in (1, 2):
            eng.forward(a, b, k / 3, tmp, scale=1.0, ensemble=False)
            want = tmp.cpu()
            got = out[3 * pair + k]
            assert (got - want).abs().max().item() <= 1e-6, describe_diff(got, want, f"pair {pair} k {k}")
        assert torch.equal(out[3 * pair], frames[pair])
    eng.release_workspace()
