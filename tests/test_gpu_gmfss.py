"""-m gpu: GMFSS Fortuna (union), SURVEY.md 8f rank 3, on the MI355X against oracle/gmfss_oracle.py (bit-exact vs the
reference's model here, oracle/VALIDATION_GMFSS.log).  Same stage-by-stage procedure as the CPU orchestration test
(tests/test_gmfss_engine_cpu.py: check_against_oracle), now with the real backend: the MFMA layer kernels on GMFSS's shapes
and the launch side of csrc/gmfss_ops.hip.  The 1e-3 gate applies to render() on the oracle's state; see the docstring of
check_against_oracle for why end-to-end agreement with random GMFlow weights is asserted statistically."""

import pytest
import torch

from cfi_amd import synth
from cfi_amd.schedule import InterpolationStateList
from oracle import gmfss_oracle as G
from test_gmfss_engine_cpu import check_against_oracle, end_to_end_gate

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["union", "base"])
def setup(request, hip_lib):
    from cfi_amd.gmfss import GMFSSEngine

    torch.cuda.set_device(0)
    sds = synth.gmfss_synth_state_dicts(1234, request.param)
    eng = GMFSSEngine(sds)
    yield sds, eng
    eng.close()


@pytest.mark.parametrize("h,w,t", [(64, 64, 0.5), (100, 150, 0.25), (200, 328, 0.5)])
def test_engine_against_oracle(setup, h, w, t):
    sds, eng = setup
    fr = synth.smooth_frames(2, h, w, seed=h, shift=2.5)
    r = check_against_oracle(eng, sds, fr, t, lambda hh, ww: torch.zeros(hh, ww, 3, device="cuda"))
    print(f"GMFSS {h}x{w} t={t}: {r}")
    eng.release_workspace()


def test_node_against_oracle_loop(hip_lib, tmp_path, monkeypatch):
    """GMFSS_Fortuna_VFI.vfi — same call as the reference's node — against the oracle's node loop"""
    import cfi_amd.ckpt as K
    import cfi_amd.gmfss as M

    sds = synth.gmfss_synth_state_dicts(1234)
    paths = {}
    for part, (_, name) in M.CKPTS_PATH_CONFIG["GMFSS_fortuna_union"].items():
        paths[name] = str(tmp_path / name)
        torch.save(sds[part], paths[name])
    monkeypatch.setattr(K, "load_file_from_github_release", lambda model_type, ckpt_name: paths[ckpt_name])
    frames = synth.smooth_frames(3, 72, 100, seed=11, shift=3.0)
    before = frames.clone()
    states = InterpolationStateList([1], True)
    (out,) = M.GMFSS_Fortuna_VFI().vfi("GMFSS_fortuna_union", frames, multiplier=3, optional_interpolation_states=states)
    want = G.gmfss_vfi(sds, frames, 3, states)
    assert torch.equal(frames, before) and out.dtype == torch.float32 and out.device.type == "cpu" and out.shape == want.shape == (5, 72, 100, 3)
    assert torch.equal(out[0], frames[0]) and torch.equal(out[3], frames[1]) and torch.equal(out[4], frames[2])
    d = (out - want).abs()
    assert d.mean().item() <= 3e-3 and (d > 2e-2).float().mean().item() <= 0.05, f"max {d.max().item()} mean {d.mean().item()}"
    with pytest.raises(KeyError):
        M.GMFSS_Fortuna_VFI().vfi("GMFSS_fortuna_v2", frames)


# ---- the hard end-to-end gate on the coherent test vector -----------------------------------------------------------
@pytest.fixture(scope="module", params=["union", "base"])
def coherent(request, hip_lib):
    from cfi_amd.gmfss import GMFSSEngine

    torch.cuda.set_device(0)
    sds = synth.gmfss_coherent_state_dicts(1234, request.param)
    eng = GMFSSEngine(sds)
    yield sds, eng
    eng.close()


@pytest.mark.parametrize("h,w,seed,t", [(128, 192, 3, 0.5), (320, 512, 5, 0.5), (320, 512, 7, 1 / 3), (448, 704, 2, 2 / 3), (448, 704, 6, 0.5)])
def test_end_to_end_gate(coherent, h, w, seed, t):
    sds, eng = coherent
    mx, mean = end_to_end_gate(eng, sds, synth.texture_frames(4, h, w, seed=seed)[:2].contiguous(), t, torch.zeros(h, w, 3, device="cuda"))
    print(f"GMFSS coherent {h}x{w} seed {seed} t={t:.3f}: e2e max {mx:.2e} mean {mean:.2e}")
    eng.release_workspace()


def test_node_end_to_end_gate(hip_lib, tmp_path, monkeypatch):
    """GMFSS_Fortuna_VFI.vfi at 320x512 (>= 270x480, no padding), x3 with a skipped pair: every new frame within 1e-3 of the
    oracle's node loop, pass-through frames bit-exact."""
    import cfi_amd.ckpt as K
    import cfi_amd.gmfss as M

    sds = synth.gmfss_coherent_state_dicts(1234)
    paths = {}
    for part, (_, name) in M.CKPTS_PATH_CONFIG["GMFSS_fortuna_union"].items():
        paths[name] = str(tmp_path / name)
        torch.save(sds[part], paths[name])
    monkeypatch.setattr(K, "load_file_from_github_release", lambda model_type, ckpt_name: paths[ckpt_name])
    frames = synth.texture_frames(4, 320, 512, seed=6)       # every (pair, t) of this clip passes oracle_conditioning, see there
    states = InterpolationStateList([1], True)
    (out,) = M.GMFSS_Fortuna_VFI().vfi("GMFSS_fortuna_union", frames, multiplier=3, optional_interpolation_states=states)
    want = G.gmfss_vfi(sds, frames, 3, states)
    assert out.shape == want.shape == (8, 320, 512, 3) and out.dtype == torch.float32 and out.device.type == "cpu"
    d = (out - want).abs()
    assert d.max().item() <= 1e-3, f"GMFSS node end to end: max {d.max().item()} mean {d.mean().item()}"
    for i, j in ((0, 0), (3, 1), (4, 2), (7, 3)):
        assert torch.equal(out[i], frames[j])


# ---- 1080p: the hard gate at the headline resolution (VERDICT r3 "missing" 1) ---------------------------------------------------------
# texture_frames(cell=32) — the vector of the gates above — is ill-conditioned for the ORACLE at 1080x1920: GMFlow's global matching
# then chooses among 32 400 candidates per pixel, near-ties appear, and 2e-6 of input noise moves 6.4 % of the oracle's own pixels by
# more than 2e-4 (34 707 by more than 1e-3; seeds 1-3, also with the matching gain doubled).  A finer texture (cell=16: every 8x8
# matching patch spans its own colour gradients) removes the ties: oracle sensitivity max 6.2e-5, no pixel over 2e-4
# (oracle_conditioning asserts it on every run).  Same coherent checkpoint, the full model at full resolution: 8 soft splats per frame
# (GMFSS_Fortuna_union_arch.py:1809-1848), the 32 400-token global attention, GridNet at 1080p.
_V1080 = {}


def _vector_1080():
    if not _V1080:
        _V1080["frames"] = synth.texture_frames(4, 1080, 1920, seed=2, cell=16)[:2].contiguous()
    return _V1080["frames"]


def test_end_to_end_gate_1080p(hip_lib, oracle_threads):
    """(union variant only: the base variant runs the same kernels, its 1080p pass would cost two more oracle runs on the host)"""
    from cfi_amd.gmfss import GMFSSEngine

    sds = synth.gmfss_coherent_state_dicts(1234, "union")
    eng = GMFSSEngine(sds)
    fr = _vector_1080()
    mx, mean = end_to_end_gate(eng, sds, fr, 0.5, torch.zeros(1080, 1920, 3, device="cuda"))
    print(f"GMFSS coherent 1080x1920 (texture cell 16, seed 2) t=0.5: e2e max {mx:.2e} mean {mean:.2e}")
    eng.close()


def test_node_end_to_end_gate_1080p(hip_lib, tmp_path, monkeypatch, oracle_threads):
    """GMFSS_Fortuna_VFI.vfi — the reference node's own call — on a 1080x1920 pair: the new frame within 1e-3 of the oracle's node
    loop on EVERY pixel, pass-through frames bit-exact."""
    import cfi_amd.ckpt as K
    import cfi_amd.gmfss as M
    from test_gmfss_engine_cpu import oracle_conditioning

    sds = synth.gmfss_coherent_state_dicts(1234)
    paths = {}
    for part, (_, name) in M.CKPTS_PATH_CONFIG["GMFSS_fortuna_union"].items():
        paths[name] = str(tmp_path / name)
        torch.save(sds[part], paths[name])
    monkeypatch.setattr(K, "load_file_from_github_release", lambda model_type, ckpt_name: paths[ckpt_name])
    frames = _vector_1080()
    cond, want_mid = oracle_conditioning(sds, frames, 0.5)
    assert cond <= 2e-4, f"the 1080p vector is ill-conditioned for the oracle itself ({cond:.1e})"
    (out,) = M.GMFSS_Fortuna_VFI().vfi("GMFSS_fortuna_union", frames, multiplier=2)
    assert out.shape == (3, 1080, 1920, 3) and out.dtype == torch.float32 and out.device.type == "cpu"
    assert torch.equal(out[0], frames[0]) and torch.equal(out[2], frames[1])
    want = G.gmfss_vfi(sds, frames, 2, None)[1]
    # (the node loop hands the oracle a permuted, non-contiguous view: torch's CPU convolutions then take another path and the two
    # oracle frames differ by rounding noise — 7.6e-5 here, itself a measure of the computation's conditioning)
    assert (want - want_mid).abs().max().item() <= 2e-4
    d = (out[1] - want).abs()
    print(f"GMFSS node 1080x1920 x2: max|d| {d.max().item():.2e} mean {d.mean().item():.2e} (oracle conditioning {cond:.1e})")
    assert d.max().item() <= 1e-3, f"GMFSS node end to end @1080p: max {d.max().item()} mean {d.mean().item()}"
