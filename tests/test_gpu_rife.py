"""-m gpu: RIFE 4.7 hot path on the MI355X vs the oracle (and vs outputs of the real reference in
tests/golden).  Contract (BASELINE.json north_star): per-pixel fp32 |d| <= 1e-3."""
import os

import numpy as np
import pytest
import torch

from gpu_util import describe_diff
from cfi_amd import synth
from cfi_amd.schedule import InterpolationStateList
from oracle import rife_oracle

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def sd():
    return synth.rife47_synth_state_dict(1234)


@pytest.fixture(scope="module")
def engine(hip_lib, sd):
    from cfi_amd.rife import RifeEngine

    torch.cuda.set_device(0)
    e = RifeEngine(sd, "4.7")
    yield e
    e.close()


def _oracle_mid(sd, frames, tasks):
    x = frames[..., :3].permute(0, 3, 1, 2)
    f0 = torch.cat([x[p:p + 1] for p, _ in tasks]).float()
    f1 = torch.cat([x[p + 1:p + 2] for p, _ in tasks]).float()
    ts = torch.tensor([t for _, t in tasks]).view(-1, 1, 1, 1)
    with torch.inference_mode():
        out, aux = rife_oracle.ifnet47_forward(sd, f0, f1, ts, (8, 4, 2, 1), return_aux=True)
    return out.clamp(0, 1).permute(0, 2, 3, 1).contiguous(), aux


@pytest.mark.parametrize("h,w", [(64, 64), (100, 150), (270, 480)])
def test_stage_by_stage_against_oracle(engine, sd, h, w):
    """Localises a failure: frame packs, every stage's flow, then the output."""
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(2, h, w, seed=h, shift=2.5)
    tasks = [(0, 0.5), (0, 0.25)]
    engine.debug_keep(True)
    try:
        got = run_tasks(engine, frames, tasks, batch_size=2)
        hp, wp = -(-h // 64) * 64, -(-w // 64) * 64
        want, aux = _oracle_mid(sd, frames, tasks)
        # encode / frame pack of the first loaded frame
        x = frames[..., :3].permute(0, 3, 1, 2)
        msgs = []
        ok = True
        for st in range(4):
            fl = engine.debug_read(0, st, 2 * hp * wp * 4).view(2, hp, wp, 4)
            wf = aux[st][0].permute(0, 2, 3, 1)
            if st == 3:  # the last block is fused with the output kernel, which only visits the un-padded HxW
                fl, wf = fl[:, :h, :w], wf[:, :h, :w]
            d = (fl - wf).abs().max().item()
            msgs.append(describe_diff(fl, wf, f"flow after block{st}"))
            ok &= d <= 2e-3
        d = (got - want).abs().max().item()
        msgs.append(describe_diff(got, want, "output"))
        assert ok and d <= TOL, "\n".join(msgs)
    finally:
        engine.debug_keep(False)


def test_frame_pack_against_oracle(engine, sd):
    """prep (clamp/pad) + encode conv/deconv vs oracle.encode."""
    from cfi_amd.rife import run_tasks

    h, w = 70, 90
    frames = synth.smooth_frames(2, h, w, seed=1, shift=1.0) * 1.2 - 0.1   # exercises the clamp
    run_tasks(engine, frames, [(0, 0.5)], batch_size=1)
    hp, wp = 128, 128
    img = torch.nn.functional.pad(frames[..., :3].permute(0, 3, 1, 2).clamp(0, 1), (0, wp - w, 0, hp - h))
    with torch.inference_mode():
        feat = rife_oracle.encode(sd, img)
    found = 0
    for slot in range(engine.cfg[3]):      # every frame slot of the configuration (run_tasks packs into whichever is free)
        pk = engine.debug_read(2, slot, hp * wp * 8).view(2, hp, wp, 4)   # planar4: [rgb0 | features]
        for k in range(2):
            if (pk[0][..., :3] - img[k].permute(1, 2, 0)).abs().max().item() == 0.0:
                found += 1
                d = (pk[1] - feat[k].permute(1, 2, 0)).abs().max().item()
                assert d <= 2e-5, describe_diff(pk[1], feat[k].permute(1, 2, 0), "encode features")
                assert pk[0][..., 3].abs().max().item() == 0.0
    assert found == 2, "frame packs not found / image channels not bit-exact"


def test_against_reference_golden(engine, sd, golden_dir):
    from cfi_amd.rife import run_tasks

    g = np.load(os.path.join(golden_dir, "rife47_net_anime.npz"))
    frames = torch.from_numpy(g["frames"])
    tasks = [(0, float(t)) for t in g["timesteps"]]
    got = run_tasks(engine, frames, tasks, batch_size=2)
    want = torch.from_numpy(g["out"]).clamp(0, 1)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, "vs reference golden")


def test_batch_invariance_and_determinism(engine, sd):
    """Per-task results do not depend on how tasks are batched (the reference: bit-identical)."""
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(4, 128, 192, seed=4, shift=3.0)
    tasks = [(0, 0.5), (1, 0.5), (2, 0.5), (1, 0.25), (0, 0.75)]
    a = run_tasks(engine, frames, tasks, batch_size=1)
    b = run_tasks(engine, frames, tasks, batch_size=4)
    c = run_tasks(engine, frames, tasks, batch_size=4)
    assert torch.equal(b, c), "run-to-run non-determinism"
    # bit-equal: every kernel of the network processes a task independently of its launch mates, and the tile variant (= fp32
    # summation order) of a layer is chosen from the image size, never from the batch (conv_pick_variant; the Winograd ResConvs work
    # per 16x8-pixel region)
    assert torch.equal(a, b), describe_diff(a, b, "batch 1 vs batch 4")
    big = synth.smooth_frames(2, 544, 960, seed=6, shift=3.0)          # sizes where the old per-launch heuristics switched variants
    t2 = [(0, 0.5), (0, 0.25), (0, 0.75)]
    assert torch.equal(run_tasks(engine, big, t2, batch_size=1), run_tasks(engine, big, t2, batch_size=3)), "batch 1 vs 3 @544x960"


def test_scale_factor_half(engine, sd):
    """scale_factor=0.5 -> block scales [16,8,4,2]: exercises the un-fused stage kernels (16->8) and the
    bilinear up-resize inside the output kernel (last scale 2)."""
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(2, 120, 200, seed=8, shift=3.0)
    tasks = [(0, 0.5), (0, 0.7)]
    got = run_tasks(engine, frames, tasks, batch_size=2, scale_factor=0.5)
    x = frames.permute(0, 3, 1, 2)
    ts = torch.tensor([0.5, 0.7]).view(-1, 1, 1, 1)
    with torch.inference_mode():
        want = rife_oracle.ifnet47_forward(sd, x[0:1].repeat(2, 1, 1, 1), x[1:2].repeat(2, 1, 1, 1), ts,
                                           (16.0, 8.0, 4.0, 2.0)).clamp(0, 1).permute(0, 2, 3, 1)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, "scale_factor 0.5")
    with pytest.raises(RuntimeError):
        run_tasks(engine, frames, tasks, batch_size=1, scale_factor=3.0)   # not one of the widget's values


@pytest.mark.parametrize("sf,h,w", [(2.0, 120, 200), (4.0, 70, 100), (2.0, 270, 480)])
def test_scale_factor_above_one(engine, sd, sf, h, w):
    """scale_factor 2 / 4 -> block scales [4,2,1,0.5] / [2,1,0.5,0.25]: the last block(s) run above the frame resolution
    (IFBlock up-samples its input by 1/scale and down-samples its output, rife_arch.py:237-276)."""
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(2, h, w, seed=8, shift=3.0)
    tasks = [(0, 0.5), (0, 0.3)]
    got = run_tasks(engine, frames, tasks, batch_size=2, scale_factor=sf)
    x = frames.permute(0, 3, 1, 2)
    ts = torch.tensor([0.5, 0.3]).view(-1, 1, 1, 1)
    with torch.inference_mode():
        want = rife_oracle.ifnet47_forward(sd, x[0:1].repeat(2, 1, 1, 1), x[1:2].repeat(2, 1, 1, 1), ts,
                                           tuple(b / sf for b in (8.0, 4.0, 2.0, 1.0))).clamp(0, 1).permute(0, 2, 3, 1)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"scale_factor {sf}")


def test_full_size_1080p(engine, sd):
    """BASELINE.json configs[1] size (padded 1088x1920), one pair, against the oracle."""
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(2, 1080, 1920, seed=2, shift=4.0)
    tasks = [(0, 0.5)]
    got = run_tasks(engine, frames, tasks, batch_size=1)
    want, _ = _oracle_mid(sd, frames, tasks)
    assert got.shape == (1, 1080, 1920, 3)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, "1080p")


def test_full_size_noise_worst_case(engine, sd):
    """i.i.d. noise frames: the worst-case-gradient input of BASELINE.md."""
    from cfi_amd.rife import run_tasks

    frames = synth.noise_frames(2, 540, 960, seed=0)
    got = run_tasks(engine, frames, [(0, 0.5)], batch_size=1)
    want, _ = _oracle_mid(sd, frames, [(0, 0.5)])
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, "noise 540p")


NODE_CASES = {
    "m2": dict(multiplier=2),
    "m3_bs2": dict(multiplier=3, batch_size=2),
    "mlist": dict(multiplier=[3, 0, 1]),
    "m2_skip12": dict(multiplier=2, optional_interpolation_states=InterpolationStateList([1, 2], True)),
    "m2_keep12": dict(multiplier=2, optional_interpolation_states=InterpolationStateList([1, 2], False)),
}


@pytest.mark.parametrize("name", list(NODE_CASES))
def test_node_against_reference_golden(hip_lib, sd, golden_dir, name, tmp_path, monkeypatch):
    """RIFE_VFI.vfi — same call as the reference's node — vs outputs of the reference node."""
    import cfi_amd.rife as R

    pth = tmp_path / "rife47.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    g = np.load(os.path.join(golden_dir, "rife47_node.npz"))
    frames = torch.from_numpy(g["frames"])
    before = frames.clone()
    (out,) = R.RIFE_VFI().vfi("rife47.pth", frames, **NODE_CASES[name])
    want = torch.from_numpy(g[name])
    assert torch.equal(frames, before), "input tensor was mutated"
    assert out.dtype == torch.float32 and out.device.type == "cpu" and out.shape == want.shape
    assert (out - want).abs().max().item() <= TOL, describe_diff(out, want, name)
    # pass-through frames are bit-exact copies of the inputs (alpha dropped)
    same = (out == want).reshape(out.shape[0], -1).all(1)
    srcs = [i for i in range(out.shape[0]) if any(torch.equal(want[i], frames[j, ..., :3]) for j in range(len(frames)))]
    assert all(bool(same[i]) for i in srcs)


def test_node_rejects_unsupported(hip_lib, sd, tmp_path, monkeypatch):
    import cfi_amd.rife as R

    with pytest.raises(KeyError):
        R.RIFE_VFI().vfi("nope.pth", torch.zeros(2, 8, 8, 3))
    with pytest.raises(NotImplementedError):
        R.RifeEngine(sd, "4.0")


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_node_dtype_widget(hip_lib, sd, tmp_path, monkeypatch, dtype):
    """dtype widget (rife/__init__.py:120-134,195-198,210,227-230,237-238): the reference rounds every frame through the
    requested dtype and ALWAYS returns float32.  Here: clip rounded to that dtype on the way in, fp32 compute, new frames
    rounded once through it, float32 out."""
    import cfi_amd.rife as R

    pth = tmp_path / "rife47.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    td = getattr(torch, dtype)
    frames = synth.smooth_frames(3, 70, 90, seed=5, shift=2.0)
    with pytest.warns(UserWarning):
        (out,) = R.RIFE_VFI().vfi("rife47.pth", frames, multiplier=2, dtype=dtype)
    (ref,) = R.RIFE_VFI().vfi("rife47.pth", frames.to(td).to(torch.float32), multiplier=2)
    assert out.dtype == torch.float32 and out.device.type == "cpu" and out.shape == ref.shape == (5, 70, 90, 3)
    out.numpy()                                                   # what SaveImage / Preview do next
    assert torch.equal(out, ref.to(td).to(torch.float32))
    assert torch.equal(out[0], frames[0].to(td).float()) and torch.equal(out[4], frames[2].to(td).float())   # pass-through frames
    # r5: an ORACLE gate of the documented semantics (the reference's own half-precision CPU run is NaN, oracle/VALIDATION_DTYPE.log, so
    # there is no reference output to pin): the fp32 oracle on the clip rounded through `td`, its frames rounded once through `td`.
    # A value within 1e-4 of the oracle's may fall on the other side of a rounding boundary: at most one ulp of `td` at 1.0
    # (fp16 2^-11 for [0.5, 1), bf16 2^-8), on few pixels; everywhere else the rounded values are EQUAL.
    from oracle import rife_oracle

    want = rife_oracle.rife_vfi(sd, frames.to(td).to(torch.float32), multiplier=2).to(td).to(torch.float32)
    d = (out - want).abs()
    ulp = 2.0 ** -11 if td == torch.float16 else 2.0 ** -8
    assert d.max().item() <= ulp * 1.0001, d.max().item()
    assert (d > 0).float().mean().item() <= 0.05, (d > 0).float().mean().item()      # measured: fp16 ~1 %, bf16 ~0.2 % of the values
    with pytest.raises(KeyError):
        R.RIFE_VFI().vfi("rife47.pth", frames, dtype="float64")


def test_config0_anime_pair_vs_reference_node(hip_lib, sd, golden_dir, tmp_path, monkeypatch):
    """BASELINE.json configs[0]: the node on the full demo pair anime0+anime1 (540x960), 2x, against the frame the
    reference node synthesised on torch-CPU (tests/golden/rife47_node_anime540.npz, oracle/make_golden.py)."""
    import cfi_amd.rife as R

    pth = tmp_path / "rife47.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    g = np.load(os.path.join(golden_dir, "rife47_node_anime540.npz"))
    frames = torch.from_numpy(g["frames_u8"].astype(np.float32) / 255.0)
    (out,) = R.RIFE_VFI().vfi("rife47.pth", frames, multiplier=2)
    assert out.shape == (3, 540, 960, 3) and torch.equal(out[0], frames[0]) and torch.equal(out[2], frames[1])
    want = torch.from_numpy(g["mid"])
    assert (out[1] - want).abs().max().item() <= TOL, describe_diff(out[1], want, "anime 540p node")


def test_config3_rife49_4k_x4(hip_lib, sd, tmp_path, monkeypatch):
    """BASELINE.json configs[3] on one rank: rife49.pth (runs as arch 4.7, rife/__init__.py:10-20), 4x multiplier on a
    4K pair (2160x3840, padded 2176x3840), through the node, against the oracle (3 forwards, ~15 s on the host)."""
    import cfi_amd.rife as R

    pth = tmp_path / "rife49.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    R._model_cache.clear()
    frames = synth.smooth_frames(2, 2160, 3840, seed=6, shift=6.0)
    (out,) = R.RIFE_VFI().vfi("rife49.pth", frames, multiplier=4, batch_size=3)
    R._model_cache.clear()
    assert out.shape == (5, 2160, 3840, 3) and torch.equal(out[0], frames[0]) and torch.equal(out[4], frames[1])
    want, _ = _oracle_mid(sd, frames, [(0, 0.25), (0, 0.5), (0, 0.75)])
    assert (out[1:4] - want).abs().max().item() <= TOL, describe_diff(out[1:4], want, "4K x4")


# ---- arch 4.17 (rife417.pth): Head_417 encoder (3 convs + transposed conv), 8 feature channels per frame -----------

@pytest.fixture(scope="module")
def sd417():
    return synth.rife417_synth_state_dict(1234)


@pytest.fixture(scope="module")
def engine417(hip_lib, sd417):
    from cfi_amd.rife import RifeEngine

    e = RifeEngine(sd417, "4.17")
    yield e
    e.close()


@pytest.mark.parametrize("h,w,sf", [(64, 64, 1.0), (100, 150, 1.0), (270, 480, 1.0), (120, 200, 0.5), (120, 200, 2.0)])
def test_rife417_against_oracle(engine417, sd417, h, w, sf):
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(2, h, w, seed=4, shift=3.0)
    tasks = [(0, 0.5), (0, 0.2)]
    got = run_tasks(engine417, frames, tasks, batch_size=2, scale_factor=sf)
    x = frames.permute(0, 3, 1, 2)
    ts = torch.tensor([0.5, 0.2]).view(-1, 1, 1, 1)
    with torch.inference_mode():
        want, aux = rife_oracle.ifnet47_forward(sd417, x[0:1].repeat(2, 1, 1, 1), x[1:2].repeat(2, 1, 1, 1), ts,
                                                tuple(b / sf for b in (8.0, 4.0, 2.0, 1.0)), return_aux=True, arch="4.17")
    want = want.clamp(0, 1).permute(0, 2, 3, 1)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"rife 4.17 {h}x{w} sf={sf}")


def test_rife417_frame_pack_and_flows(engine417, sd417):
    """debug taps: the frame pack (rgb + 8 encoder features = Head_417 output) and every stage's flow"""
    from cfi_amd.rife import run_tasks

    h, w = 100, 150
    hp, wp = 128, 192
    frames = synth.smooth_frames(2, h, w, seed=4, shift=3.0)
    engine417.debug_keep(True)
    try:
        run_tasks(engine417, frames, [(0, 0.5)], batch_size=1)
        x = frames.permute(0, 3, 1, 2)
        with torch.inference_mode():
            _, aux = rife_oracle.ifnet47_forward(sd417, x[0:1], x[1:2], torch.tensor([0.5]).view(1, 1, 1, 1), return_aux=True, arch="4.17")
            feat = rife_oracle.encode417(sd417, torch.nn.functional.pad(x[0:1].clamp(0, 1), (0, wp - w, 0, hp - h)))
        slot = None
        for s_ in range(engine417.cfg[3]):   # find the slot that holds frame 0 (slot assignment is an implementation detail)
            pack = engine417.debug_read(2, s_, 3 * hp * wp * 4).view(3, hp, wp, 4)
            if torch.equal(pack[0, :h, :w, :3], frames[0].clamp(0, 1)):
                slot = s_
                break
        assert slot is not None, "frame 0 not found in any slot"
        got_feat = torch.cat([pack[1], pack[2]], -1).permute(2, 0, 1)[None]
        assert (got_feat - feat).abs().max().item() <= 2e-5, describe_diff(got_feat, feat, "Head_417 features", chan_last=False)
        for i in range(4):
            fl = engine417.debug_read(0, i, hp * wp * 4).view(1, hp, wp, 4).permute(0, 3, 1, 2)
            wf = aux[i][0]
            sl = (slice(None), slice(None), slice(0, h), slice(0, w)) if i == 3 else (slice(None),) * 4
            assert (fl[sl] - wf[sl]).abs().max().item() <= 2e-4, describe_diff(fl[sl], wf[sl], f"flow after block {i}", chan_last=False)
    finally:
        engine417.debug_keep(False)


def test_rife417_1080p(engine417, sd417):
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(2, 1080, 1920, seed=2, shift=4.0)
    got = run_tasks(engine417, frames, [(0, 0.5)], batch_size=1)
    x = frames.permute(0, 3, 1, 2)
    with torch.inference_mode():
        want = rife_oracle.ifnet47_forward(sd417, x[0:1], x[1:2], torch.tensor([0.5]).view(1, 1, 1, 1), arch="4.17")
    want = want.clamp(0, 1).permute(0, 2, 3, 1)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, "rife 4.17 1080p")


@pytest.mark.parametrize("name,kw", [("m2", dict(multiplier=2)), ("mlist_bs2", dict(multiplier=[3, 1], batch_size=2))])
def test_node417_against_reference_golden(hip_lib, sd417, golden_dir, tmp_path, monkeypatch, name, kw):
    """RIFE_VFI.vfi("rife417.pth", ...) vs the reference node's own output (tests/golden/rife417_node.npz)"""
    import cfi_amd.rife as R

    pth = tmp_path / "rife417.pth"
    torch.save(sd417, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    R._model_cache.clear()
    g = np.load(os.path.join(golden_dir, "rife417_node.npz"))
    (out,) = R.RIFE_VFI().vfi("rife417.pth", torch.from_numpy(g["frames"]), **kw)
    R._model_cache.clear()
    want = torch.from_numpy(g[name])
    assert out.shape == want.shape and (out - want).abs().max().item() <= TOL, describe_diff(out, want, name)


# ---- arch 4.26 (rife426.pth): Head encoder, 5 IFBlocks at scales [16,8,4,2,1], 8 block-feature channels carried on ----

@pytest.fixture(scope="module")
def sd426():
    return synth.rife426_synth_state_dict(1234)


@pytest.fixture(scope="module")
def engine426(hip_lib, sd426):
    from cfi_amd.rife import RifeEngine

    e = RifeEngine(sd426, "4.26")
    yield e
    e.close()


@pytest.mark.parametrize("h,w,sf", [(64, 64, 1.0), (100, 150, 1.0), (270, 480, 1.0), (120, 200, 0.5), (120, 200, 2.0), (70, 100, 4.0)])
def test_rife426_against_oracle(engine426, sd426, h, w, sf):
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(2, h, w, seed=4, shift=3.0)
    tasks = [(0, 0.5), (0, 0.2)]
    got = run_tasks(engine426, frames, tasks, batch_size=2, scale_factor=sf)
    x = frames.permute(0, 3, 1, 2)
    ts = torch.tensor([0.5, 0.2]).view(-1, 1, 1, 1)
    with torch.inference_mode():
        want = rife_oracle.ifnet47_forward(sd426, x[0:1].repeat(2, 1, 1, 1), x[1:2].repeat(2, 1, 1, 1), ts,
                                           tuple(b / sf for b in (16.0, 8.0, 4.0, 2.0, 1.0)), arch="4.26")
    want = want.clamp(0, 1).permute(0, 2, 3, 1)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"rife 4.26 {h}x{w} sf={sf}")


def test_rife426_flows_per_block(engine426, sd426):
    """debug taps: flow after each of the 5 blocks (localises a failure)"""
    from cfi_amd.rife import run_tasks

    h, w, hp, wp = 100, 150, 128, 192
    frames = synth.smooth_frames(2, h, w, seed=4, shift=3.0)
    engine426.debug_keep(True)
    try:
        run_tasks(engine426, frames, [(0, 0.5)], batch_size=1)
        x = frames.permute(0, 3, 1, 2)
        with torch.inference_mode():
            _, aux = rife_oracle.ifnet47_forward(sd426, x[0:1], x[1:2], torch.tensor([0.5]).view(1, 1, 1, 1), (16, 8, 4, 2, 1),
                                                 return_aux=True, arch="4.26")
        for i in range(5):
            fl = engine426.debug_read(0, i, hp * wp * 4).view(1, hp, wp, 4).permute(0, 3, 1, 2)
            wf = aux[i][0]
            sl = (slice(None), slice(None), slice(0, h), slice(0, w)) if i == 4 else (slice(None),) * 4
            assert (fl[sl] - wf[sl]).abs().max().item() <= 2e-4, describe_diff(fl[sl], wf[sl], f"flow after block {i}", chan_last=False)
    finally:
        engine426.debug_keep(False)


def test_rife426_1080p(engine426, sd426):
    from cfi_amd.rife import run_tasks

    frames = synth.smooth_frames(2, 1080, 1920, seed=2, shift=4.0)
    got = run_tasks(engine426, frames, [(0, 0.5)], batch_size=1)
    x = frames.permute(0, 3, 1, 2)
    with torch.inference_mode():
        want = rife_oracle.ifnet47_forward(sd426, x[0:1], x[1:2], torch.tensor([0.5]).view(1, 1, 1, 1), (16, 8, 4, 2, 1), arch="4.26")
    want = want.clamp(0, 1).permute(0, 2, 3, 1)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, "rife 4.26 1080p")


@pytest.mark.parametrize("name,kw", [("m2", dict(multiplier=2)), ("mlist_bs2", dict(multiplier=[3, 1], batch_size=2))])
def test_node426_against_reference_golden(hip_lib, sd426, golden_dir, tmp_path, monkeypatch, name, kw):
    """RIFE_VFI.vfi("rife426.pth", ...) vs the reference node's own output (tests/golden/rife426_node.npz)"""
    import cfi_amd.rife as R

    pth = tmp_path / "rife426.pth"
    torch.save(sd426, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    R._model_cache.clear()
    g = np.load(os.path.join(golden_dir, "rife426_node.npz"))
    (out,) = R.RIFE_VFI().vfi("rife426.pth", torch.from_numpy(g["frames"]), **kw)
    R._model_cache.clear()
    want = torch.from_numpy(g[name])
    assert out.shape == want.shape and (out - want).abs().max().item() <= TOL, describe_diff(out, want, name)


# ---- edge cases of the node contract ---------------------------------------------------------------------------------

def _node(sd, tmp_path, monkeypatch):
    import cfi_amd.rife as R

    pth = tmp_path / "rife47.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    R._model_cache.clear()
    return R


def test_node_edge_no_new_frames(hip_lib, sd, tmp_path, monkeypatch):
    """single-frame clip, multiplier 1, and every pair skipped: nothing to synthesise, frames pass through bit-exact"""
    R = _node(sd, tmp_path, monkeypatch)
    fr = synth.smooth_frames(3, 40, 56, seed=1)
    (o,) = R.RIFE_VFI().vfi("rife47.pth", fr[:1], multiplier=2)
    assert torch.equal(o, fr[:1])
    (o,) = R.RIFE_VFI().vfi("rife47.pth", fr, multiplier=1)
    assert torch.equal(o, fr)
    (o,) = R.RIFE_VFI().vfi("rife47.pth", fr, multiplier=3, optional_interpolation_states=InterpolationStateList([0, 1], True))
    assert torch.equal(o, fr)
    R._model_cache.clear()


def test_node_edge_tiny_and_strided_input(hip_lib, sd, tmp_path, monkeypatch):
    """5x7 frames (padded to 64x64 inside) and a non-contiguous RGBA input view; the input must not be modified"""
    R = _node(sd, tmp_path, monkeypatch)
    tiny = synth.noise_frames(2, 5, 7, seed=3)
    (o,) = R.RIFE_VFI().vfi("rife47.pth", tiny, multiplier=2)
    want = rife_oracle.rife_vfi(sd, tiny, multiplier=2)
    assert o.shape == (3, 5, 7, 3) and (o - want).abs().max().item() <= TOL
    big = synth.smooth_frames(6, 48, 80, seed=2, c=4)
    view = big[::2, 4:44, 8:72]            # strided in N, H and W, 4 channels
    before = big.clone()
    (o,) = R.RIFE_VFI().vfi("rife47.pth", view, multiplier=2, batch_size=4)
    want = rife_oracle.rife_vfi(sd, view, multiplier=2)
    assert torch.equal(big, before)
    assert o.shape == want.shape and (o - want).abs().max().item() <= TOL, describe_diff(o, want, "strided input")
    assert torch.equal(o[0], view[0, ..., :3]) and torch.equal(o[-1], view[-1, ..., :3])
    R._model_cache.clear()


def test_engine_rejects_bad_calls(engine):
    from cfi_amd import _lib

    with pytest.raises(RuntimeError):
        engine.configure(64, 64, 1, 4, 3.0)            # scale_factor outside the widget's values
    with pytest.raises(RuntimeError):
        engine.configure(64, 64, 64, 4, 1.0)           # batch above the library's task table
    engine.configure(64, 64, 2, 4, 1.0)
    out = torch.empty(3, 64, 64, 3, device="cuda")
    with pytest.raises(RuntimeError):
        engine.interpolate([0, 1, 2], [1, 2, 3], [0.5] * 3, out)   # batch above the configured maximum
    with pytest.raises(RuntimeError):
        engine.interpolate([0], [9], [0.5], out[:1])               # slot outside the frame cache
    assert "slot" in _lib.last_error() or "batch" in _lib.last_error()


def test_large_flows(hip_lib):
    """Real checkpoints move pixels by tens of px; the synthetic default moves them by ~3.  Scale every lastconv x8
    (flows and mask logits of tens of units, warps far outside the frame at the borders) and check parity still holds."""
    from cfi_amd.rife import RifeEngine, run_tasks

    sd = synth.rife47_synth_state_dict(77)
    big = {k: (v * 8.0 if "lastconv" in k else v) for k, v in sd.items()}
    eng = RifeEngine(big, "4.7")
    try:
        frames = synth.smooth_frames(2, 200, 328, seed=12, shift=6.0)
        tasks = [(0, 0.5), (0, 0.125)]
        got = run_tasks(eng, frames, tasks, batch_size=2)
        want, aux = _oracle_mid(big, frames, tasks)
        fmax = max(a[0].abs().max().item() for a in aux)
        assert fmax > 15.0, f"test premise: large flows (got {fmax:.1f} px)"
        assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"large flows ({fmax:.0f} px)")
    finally:
        eng.close()


# ---- arch 4.0 (sudo_rife4 checkpoint): op-by-op engine (rife40.py), the node's fast_mode / ensemble widgets matter here ----

CKPT40 = "sudo_rife4_269.662_testV1_scale1.pth"


@pytest.fixture(scope="module")
def sd40():
    return synth.rife40_synth_state_dict(1234)


def _run40(sd, frames, ts, scale_list, training, fastmode):
    from cfi_amd.rife40 import Rife40Engine

    eng = Rife40Engine(sd)
    try:
        h, w = frames.shape[1:3]
        eng.configure(h, w, len(ts))
        f0, f1 = frames[0].cuda().contiguous(), frames[1].cuda().contiguous()
        out = torch.empty(len(ts), h, w, 3, device="cuda")
        eng.forward([f0] * len(ts), [f1] * len(ts), ts, scale_list, training, fastmode, out)
        return out.cpu()
    finally:
        eng.close()


@pytest.mark.parametrize("h,w,training,fastmode,sf", [(64, 64, True, True, 1.0), (100, 150, False, False, 1.0), (270, 480, True, False, 1.0),
                                                      (120, 200, False, True, 0.5), (100, 150, False, False, 2.0)])
def test_rife40_against_oracle(hip_lib, sd40, h, w, training, fastmode, sf):
    frames = synth.smooth_frames(2, h, w, seed=4, shift=3.0)
    ts = [0.5, 0.2]
    got = _run40(sd40, frames, ts, [8 / sf, 4 / sf, 2 / sf, 1 / sf], training, fastmode)
    x = frames.permute(0, 3, 1, 2)
    with torch.inference_mode():
        want = rife_oracle.ifnet40_forward(sd40, x[0:1].repeat(2, 1, 1, 1), x[1:2].repeat(2, 1, 1, 1), torch.tensor(ts).view(-1, 1, 1, 1),
                                           [8 / sf, 4 / sf, 2 / sf, 1 / sf], training, fastmode)
    want = want.clamp(0, 1).permute(0, 2, 3, 1)
    assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"rife 4.0 {h}x{w} training={training} fastmode={fastmode} sf={sf}")


def test_rife40_scale_doubling(hip_lib, sd40):
    """block-1 flows above 32 px in both directions with training=False: the block scales are doubled in place and blocks 0/1
    re-run (rife_arch.py:598-607); with training=True the same weights must NOT double."""
    big = {k: (v * 6.0 if "lastconv" in k and k[:6] in ("block0", "block1") else v) for k, v in sd40.items()}
    frames = synth.smooth_frames(2, 128, 192, seed=3, shift=2.5)
    x = frames.permute(0, 3, 1, 2)
    for training in (False, True):
        sl_g, sl_o = [8.0, 4.0, 2.0, 1.0], [8.0, 4.0, 2.0, 1.0]
        got = _run40(big, frames, [0.5], sl_g, training, False)
        with torch.inference_mode():
            want = rife_oracle.ifnet40_forward(big, x[0:1], x[1:2], torch.tensor([0.5]).view(1, 1, 1, 1), sl_o, training, False)
        want = want.clamp(0, 1).permute(0, 2, 3, 1)
        assert sl_g == sl_o == ([16.0, 8.0, 4.0, 2.0] if not training else [8.0, 4.0, 2.0, 1.0]), (training, sl_g, sl_o)
        assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, f"rife 4.0 doubling training={training}")


@pytest.mark.parametrize("name,kw", [("default", dict(multiplier=2, fast_mode=True, ensemble=True)),
                                     ("refine_bs2", dict(multiplier=[3, 1], batch_size=2, fast_mode=False, ensemble=False))])
def test_node40_against_reference_golden(hip_lib, sd40, golden_dir, tmp_path, monkeypatch, name, kw):
    import cfi_amd.rife as R

    pth = tmp_path / CKPT40
    torch.save(sd40, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    R._model_cache.clear()
    g = np.load(os.path.join(golden_dir, "rife40_node.npz"))
    frames = torch.from_numpy(g["frames"])
    (out,) = R.RIFE_VFI().vfi(CKPT40, frames, **kw)
    for e in R._model_cache.values():
        e.close()
    R._model_cache.clear()
    want = torch.from_numpy(g[name])
    assert out.shape == want.shape and (out - want).abs().max().item() <= TOL, describe_diff(out, want, name)
    assert torch.equal(out[0], frames[0, ..., :3]) and torch.equal(out[-1], frames[-1, ..., :3])


def test_resconv_beta_edge_cases(hip_lib):
    """ResConv's `conv(x) * beta + x` is normally folded into the centre tap (w += 1/beta); a layer with any |beta| < 1e-2
    must take the explicit-residual path instead.  Negative and tiny betas, whole network against the oracle."""
    from cfi_amd.rife import RifeEngine, run_tasks

    sd = dict(synth.rife47_synth_state_dict(5))
    g = torch.Generator().manual_seed(9)
    for k in list(sd):
        if k.endswith("beta"):
            b = sd[k].clone()
            if "convblock.0" in k or "convblock.5" in k:
                b.view(-1)[::7] = 0.0                       # exact zeros: the block's output there is just lrelu(x)
                b.view(-1)[3::11] = 1e-4
            elif "convblock.2" in k:
                b = -b                                      # negative scale (still foldable)
            sd[k] = b
    eng = RifeEngine(sd, "4.7")
    try:
        frames = synth.smooth_frames(2, 96, 160, seed=13, shift=3.0)
        tasks = [(0, 0.5)]
        got = run_tasks(eng, frames, tasks, batch_size=1)
        want, _ = _oracle_mid(sd, frames, tasks)
        assert (got - want).abs().max().item() <= TOL, describe_diff(got, want, "beta edge cases")
    finally:
        eng.close()


# ---- block transitions: quad-per-cell kernel (default) vs the cell-per-thread kernel -------------------------------
_QUAD_SNIPPET = r"""
import sys, torch
sys.path.insert(0, {root!r})
from pkgload import load_package
load_package()
from cfi_amd import _lib, synth
from cfi_amd.rife import RifeEngine, run_tasks
torch.cuda.set_device(0)
_lib.use_test_build()      # the A/B switch lives in libvfi_hip_test.so only
assert _lib.load().vfi_test_set_option(b"stage_quad", {mask}) == 0
outs = []
for arch, mk in (("4.7", synth.rife47_synth_state_dict), ("4.17", synth.rife417_synth_state_dict)):
    e = RifeEngine(mk(77), arch)
    for h, w in ((70, 90), (200, 328)):
        frames = synth.smooth_frames(3, h, w, seed=h, shift=3.0)
        outs.append(run_tasks(e, frames, [(0, 0.5), (1, 0.3), (0, 0.8)], batch_size=3).cpu())
    e.close()
torch.save(outs, {out!r})
"""


def test_quad_transition_matches_cell_kernel(hip_lib, tmp_path):
    """stage_trans_quad_kernel restates stage_trans_kernel expression for expression: option stage_quad = 0 (cell kernel
    for every transition) and the default (quad kernel for the 8->4 and 4->2 transitions) must agree to rounding
    noise, including frames whose padded size leaves partial 16x4-cell tiles.  Not to the bit: HIP's __fmul_rn /
    __fadd_rn are plain operators, so hipcc picks the FMA contractions of the bilinear expressions per kernel
    (measured: max 1.5e-5 on the output, run-to-run identical for either kernel)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mask in ("0", "6"):
        out = str(tmp_path / f"quad{mask}.pt")
        subprocess.run([sys.executable, "-c", _QUAD_SNIPPET.format(root=root, out=out, mask=mask)], check=True, timeout=300)
        res[mask] = torch.load(out)
    for a, b in zip(res["0"], res["6"]):
        assert (a - b).abs().max().item() <= 5e-5, describe_diff(a, b, "quad vs cell transition")


def test_node_uint8_clip(hip_lib, sd, tmp_path, monkeypatch):
    """an 8-bit clip stays 8-bit on the host and over PCIe; conversion on the device both ways.  Equal, value
    for value, to the float32 node on frames / 255 followed by round(y * 255) — and 4x fewer bytes each way."""
    import cfi_amd.rife as R

    pth = tmp_path / "rife47.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    f32 = synth.smooth_frames(6, 90, 134, seed=8, shift=2.0, c=4)
    u8 = (f32 * 255).round().to(torch.uint8)                                 # RGBA, 6 frames
    before = u8.clone()
    (out,) = R.RIFE_VFI().vfi("rife47.pth", u8, multiplier=3, batch_size=4)
    (ref,) = R.RIFE_VFI().vfi("rife47.pth", u8.to(torch.float32) / 255.0, multiplier=3, batch_size=4)
    assert torch.equal(u8, before) and out.dtype == torch.uint8 and out.device.type == "cpu" and out.shape == (16, 90, 134, 3)
    want = (ref.clamp(0, 1) * 255).round().to(torch.uint8)
    assert torch.equal(out, want), f"{(out.int() - want.int()).abs().max().item()} levels"
    for i in range(6):
        assert torch.equal(out[3 * i], u8[i, ..., :3])


def test_whole_clip_c_entry_point(hip_lib, sd, tmp_path, monkeypatch):
    """vfi_rife_run: the node call behind ONE C entry point (host clip in, host clip out) against the Python node — list
    multipliers incl. 1 and 4, a skipped pair, RGBA input, more frames than the staging ring has slots."""
    import ctypes as C

    import cfi_amd.rife as R
    from cfi_amd import _lib

    pth = tmp_path / "rife47.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    frames = torch.cat([synth.smooth_frames(4, 100, 150, seed=s_, shift=2.0, c=4) for s_ in (1, 2, 3)]).contiguous()   # 12 RGBA frames
    mult = [2, 3, 1, 4, 2, 2, 2, 3, 2, 2, 2]
    states = InterpolationStateList([4], True)
    (want,) = R.RIFE_VFI().vfi("rife47.pth", frames, multiplier=mult, batch_size=8, optional_interpolation_states=states)
    eng = R.RifeEngine(sd, "4.7")
    try:
        m = (C.c_int * 11)(*mult)
        skip = (C.c_uint8 * 11)(*[1 if i == 4 else 0 for i in range(11)])
        n_out = C.c_int64(0)
        _lib.check(hip_lib.vfi_rife_run(eng.handle, None, 12, 100, 150, 4, m, skip, 1.0, 8, None, C.byref(n_out)), "vfi_rife_run (size)")
        assert n_out.value == want.shape[0]
        out = torch.full((n_out.value, 100, 150, 3), float("nan"))
        _lib.check(hip_lib.vfi_rife_run(eng.handle, frames.data_ptr(), 12, 100, 150, 4, m, skip, 1.0, 8, out.data_ptr(), C.byref(n_out)),
                   "vfi_rife_run")
    finally:
        eng.close()
    assert not torch.isnan(out).any()
    assert (out - want).abs().max().item() <= 2e-5, describe_diff(out, want, "vfi_rife_run vs the node")
    src = [i for i in range(want.shape[0]) if any(torch.equal(want[i], frames[j, ..., :3]) for j in range(12))]
    assert len(src) == 12 and all(torch.equal(out[i], want[i]) for i in src)


@pytest.mark.parametrize("h,w,bs", [(128, 192, 1), (270, 480, 3), (1080, 1920, 2)])
def test_fused_last_transition_matches_unfused(hip_lib, sd, h, w, bs):
    """trans1_conv0a (block transition 2 -> 1 fused into block 3's conv0.0, X never in HBM, flow ping-pong) against the
    un-fused kernels — which the debug taps force (they need X) — and against the oracle."""
    from cfi_amd.rife import RifeEngine, run_tasks

    eng = RifeEngine(sd, "4.7")
    try:
        frames = synth.smooth_frames(bs + 1, h, w, seed=h + bs, shift=3.0)
        tasks = [(p, 0.5 if p % 2 == 0 else 0.3) for p in range(bs)]
        fused = run_tasks(eng, frames, tasks, batch_size=bs)
        eng.debug_keep(True)
        unfused = run_tasks(eng, frames, tasks, batch_size=bs)
        eng.debug_keep(False)
        again = run_tasks(eng, frames, tasks, batch_size=bs)
    finally:
        eng.close()
    assert torch.equal(fused, again), "fused path not deterministic / state left behind by the un-fused run"
    # (different summation order in conv0.0; the later blocks amplify rounding noise a little: 5e-5 at 1080p)
    assert (fused - unfused).abs().max().item() <= 2e-4, describe_diff(fused, unfused, "fused vs un-fused last transition")
    if h <= 270:
        x = frames.permute(0, 3, 1, 2)
        with torch.inference_mode():
            want = torch.cat([rife_oracle.ifnet47_forward(sd, x[p:p + 1], x[p + 1:p + 2], torch.tensor([t]).view(1, 1, 1, 1)) for p, t in tasks])
        want = want.permute(0, 2, 3, 1).clamp(0, 1)
        assert (fused - want).abs().max().item() <= 1e-3, describe_diff(fused, want, "fused path vs oracle")


def _frame_pack(h, w, u8):
    """frame pack (planar4 rgb | encode features) of a seeded frame"""
    from cfi_amd.rife import RifeEngine

    sd_ = synth.rife47_synth_state_dict(1234)
    eng = RifeEngine(sd_, "4.7")
    try:
        eng.configure(h, w, 1, 2, 1.0)
        fr = synth.smooth_frames(1, h, w, seed=9, shift=1.0, c=4)[0] * 1.3 - 0.15        # RGBA, values outside [0,1]: the clamp
        if u8:
            fr = (fr.clamp(0, 1) * 255).round().to(torch.uint8)
        eng.load_frame(1, fr.cuda().contiguous())
        hp, wp = -(-h // 64) * 64, -(-w // 64) * 64
        return eng.debug_read(2, 1, hp * wp * 8).numpy()
    finally:
        eng.close()


@pytest.mark.parametrize("h,w,u8,n", [(70, 90, False, 3), (1080, 1920, False, 5), (200, 330, True, 7), (96, 64, False, 70)])
def test_batched_frame_pack_is_bit_identical(hip_lib, h, w, u8, n):
    """vfi_rife_load_frames: n frames packed by ONE persistent launch (encode47_batch_kernel: the next tile's source prefetched into
    registers under the current tile's arithmetic; 70 frames = two launches of the 64-entry argument array) == n single-frame packs,
    bit for bit, RGBA and uint8 clips included; and A/B option encode_batched = 0 routes the same call through the single launches."""
    from cfi_amd.rife import RifeEngine

    eng = RifeEngine(synth.rife47_synth_state_dict(1234), "4.7")
    try:
        eng.configure(h, w, 1, n, 1.0)
        fr = synth.smooth_frames(n, h, w, seed=9, shift=1.0, c=4) * 1.3 - 0.15
        if u8:
            fr = (fr.clamp(0, 1) * 255).round().to(torch.uint8)
        dev = [f.cuda().contiguous() for f in fr]
        hp, wp = -(-h // 64) * 64, -(-w // 64) * 64
        slots = list(range(n))[::-1]          # not the identity
        for s, f in zip(slots, dev):
            eng.load_frame(s, f)
        want = [eng.debug_read(2, s, hp * wp * 8).clone() for s in slots]
        for mode in (1, 0):
            for s in slots:                   # poison the slots: the batched call must rewrite every one of them
                eng.load_frame(s, torch.zeros_like(dev[0]))
            assert hip_lib.vfi_test_set_option(b"encode_batched", mode) == 0
            eng.load_frames(slots, dev)
            for s, w_ in zip(slots, want):
                got = eng.debug_read(2, s, hp * wp * 8)
                assert torch.equal(got, w_), f"encode_batched={mode} slot {s}: {(got - w_).abs().max().item()}"
        assert float(want[0].abs().max()) > 0.1
        with pytest.raises(RuntimeError, match="listed twice"):
            eng.load_frames([0, 0], dev[:2])
    finally:
        hip_lib.vfi_test_set_option(b"encode_batched", 1)
        eng.close()


@pytest.mark.parametrize("h,w,u8", [(70, 90, False), (1080, 1920, False), (200, 330, True)])
def test_fused_frame_pack_is_bit_identical(hip_lib, h, w, u8):
    """arch 4.7: prep + encode.0 + encode.1 in one launch (encode47_fused_kernel, E never in HBM) gives BIT-IDENTICAL frame packs
    to the three-kernel path (A/B option fuse_encode = 0, include/vfi_hip_test.h: vfi_test_set_option)."""
    outs = []
    try:
        for flag in (1, 0):
            assert hip_lib.vfi_test_set_option(b"fuse_encode", flag) == 0
            outs.append(_frame_pack(h, w, u8))
    finally:
        hip_lib.vfi_test_set_option(b"fuse_encode", 1)
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1]), float(np.abs(outs[0] - outs[1]).max())
    assert np.abs(outs[0]).max() > 0.1


def test_reserved_compute_units_change_no_bit():
    """vfi_set_reserved_cus sizes the persistent kernels' grids for fewer compute units (room for an overlapped collective's kernel);
    a frame's bits do not depend on it."""
    from cfi_amd import _lib, synth
    from cfi_amd.rife import RifeEngine

    lib = _lib.load()
    eng = RifeEngine(synth.rife47_synth_state_dict(7), "4.7")
    H, W, B = 270, 480, 3
    eng.configure(H, W, B, B + 1, 1.0)
    g = torch.Generator().manual_seed(5)
    raw = torch.rand((B + 1, H, W, 3), generator=g).cuda()
    outs = []
    try:
        for r in (0, 32, 200, 100000):
            rc = lib.vfi_set_reserved_cus(r)
            if r > 1024:
                assert rc != 0
                continue
            assert rc == 0 and lib.vfi_get_reserved_cus() == r
            out = torch.empty((B, H, W, 3), device="cuda")
            eng.load_frames(list(range(B + 1)), [raw[j] for j in range(B + 1)])
            eng.interpolate(list(range(B)), list(range(1, B + 1)), [0.5, 0.25, 0.75], out)
            outs.append(out.cpu())
    finally:
        lib.vfi_set_reserved_cus(0)
        eng.close()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
