"""-m gpu: the three node classes on the REAL 1080p demo pair (demo_frames/bocchi0.jpg + bocchi1.jpg of the reference, SURVEY.md
§2 row 25 / §8d config 2(ii)) against what the REAL reference nodes produced for it on torch-CPU (oracle/make_golden_bocchi.py,
oracle/VALIDATION_BOCCHI.log), default synthetic checkpoints and the "hot" ones (flows of 40-100 px, occlusions).

The goldens are fingerprints (oracle/golden_stats.py): per-pixel |d| <= 1e-3 on 12 full-precision 128x128 crops (corners,
borders, centre, interior) and the mean / max of every 8x8 block of the whole frame."""
import os

import numpy as np
import pytest
import torch

from cfi_amd import synth
from oracle import golden_stats

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames(golden_dir):
    u8 = np.load(os.path.join(golden_dir, "bocchi_pair_u8.npz"))["frames_u8"]
    assert u8.shape == (2, 1080, 1920, 3)
    return torch.from_numpy(u8.astype(np.float32) / 255.0)


def _fp(npz, key):
    return {f: npz[f"{key}/{f}"] for f in ("crops", "crop_pos", "pool_mean", "pool_max")}


def _check(out, frames, m, k, npz, key, name, tol=1e-3, pool_mean_tol=golden_stats.POOL_MEAN_TOL):
    assert out.shape == (m + 1, 1080, 1920, 3) and out.dtype == torch.float32 and out.device.type == "cpu"
    assert torch.equal(out[0], frames[0]) and torch.equal(out[-1], frames[1])
    d = golden_stats.check(out[k].numpy(), _fp(npz, key), tol=tol, name=name, pool_mean_tol=pool_mean_tol)
    print(f"{name}: crops max|d| {d[0]:.2e}, 8x8 block mean max|d| {d[1]:.2e}, block max max|d| {d[2]:.2e}")


RIFE_CASES = [("default", 2, 1), ("hot", 2, 1), ("hot", 4, 1)]


@pytest.mark.parametrize("tag,m,k", RIFE_CASES)
def test_rife47_bocchi_1080p_vs_reference_node(hip_lib, frames, golden_dir, tmp_path, monkeypatch, tag, m, k):
    import cfi_amd.rife as R

    sd = synth.rife47_synth_state_dict(1234) if tag == "default" else synth.rife47_hot_state_dict(1234)
    pth = tmp_path / "rife47.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(R, "load_file_from_github_release", lambda model_type, ckpt: str(pth))
    for e in R._model_cache.values():
        e.close()
    R._model_cache.clear()
    npz = np.load(os.path.join(golden_dir, "rife47_bocchi1080.npz"))
    (out,) = R.RIFE_VFI().vfi("rife47.pth", frames, multiplier=m)
    _check(out, frames, m, k, npz, f"{tag}_x{m}_{k}", f"RIFE 4.7 {tag} x{m} frame {k} @1080p bocchi")
    for e in R._model_cache.values():
        e.close()
    R._model_cache.clear()


@pytest.mark.parametrize("tag", ["default", "hot"])
def test_film_bocchi_1080p_vs_reference_node(hip_lib, frames, golden_dir, tmp_path, monkeypatch, tag):
    """The golden is the UNMODIFIED reference node running a TorchScript trace of film_arch.Interpolator (its own loader path); the
    TorchScript executor itself is 3e-5 (default) / 5e-5 (hot) away from the eager module (oracle/VALIDATION_BOCCHI.log)."""
    import cfi_amd.film as FM
    from cfi_amd import ckpt

    sd = synth.film_synth_state_dict(1234) if tag == "default" else synth.film_hot_state_dict(1234)
    pth = tmp_path / "film_net_fp32.pt"
    torch.save(sd, pth)
    monkeypatch.setattr(FM, "load_file_from_github_release", lambda model_type, ckpt_: str(pth))
    ckpt.clear_engine_cache()
    npz = np.load(os.path.join(golden_dir, "film_bocchi1080.npz"))
    (out,) = FM.FILM_VFI().vfi("film_net_fp32.pt", frames, multiplier=2)
    ckpt.clear_engine_cache()
    # block-mean gate 1e-4: the reference's own TorchScript-vs-eager gap on this frame is 3e-5 / 5e-5 EVERYWHERE (coherent rounding
    # noise, not isolated pixels), which the single-pixel-sized default (1.8e-5) cannot separate from; measured here 2.2e-5 (hot)
    _check(out, frames, 2, 1, npz, f"{tag}_x2_1", f"FILM {tag} x2 @1080p bocchi", pool_mean_tol=1e-4)


@pytest.mark.parametrize("tag,m,k", [("default", 2, 1), ("hot", 2, 1), ("hot", 3, 1)])
def test_m2m_bocchi_1080p_vs_reference_node(hip_lib, frames, golden_dir, tmp_path, monkeypatch, tag, m, k):
    """Hot checkpoint: refined multi-branch flows up to 107 px — the summation splat is discontinuous in the flow (a source moves
    to the next target cell when its flow crosses an integer), so isolated pixels can exceed the per-pixel gate for ANY change of
    rounding; they are bounded in number instead (<= 20 per frame) and in mean."""
    from cfi_amd import ckpt, m2m

    sd = synth.m2m_synth_state_dict(1234) if tag == "default" else synth.m2m_hot_state_dict(1234)
    pth = tmp_path / "M2M.pth"
    torch.save(sd, pth)
    monkeypatch.setattr(m2m, "load_file_from_github_release", lambda model_type, ckpt_: str(pth))
    ckpt.clear_engine_cache()
    npz = np.load(os.path.join(golden_dir, "m2m_bocchi1080.npz"))
    (out,) = m2m.M2M_VFI().vfi("M2M.pth", frames, multiplier=m)
    ckpt.clear_engine_cache()
    assert out.shape == (m + 1, 1080, 1920, 3) and torch.equal(out[0], frames[0]) and torch.equal(out[-1], frames[1])
    key = f"{tag}_x{m}_{k}"
    if tag == "default":
        _check(out, frames, m, k, npz, key, f"M2M {tag} x{m} frame {k} @1080p bocchi")
        return
    fp = _fp(npz, key)
    got = golden_stats.fingerprint(out[k].numpy())
    d = np.abs(got["crops"] - fp["crops"])
    dm = np.abs(got["pool_mean"].astype(np.float64) - fp["pool_mean"])
    print(f"M2M hot x{m}: crops max|d| {d.max():.2e} mean {d.mean():.2e} n>1e-3 {(d > 1e-3).sum()}; block-mean max|d| {dm.max():.2e}")
    assert d.mean() <= 2e-6 and (d > 1e-3).sum() <= 20, (d.max(), d.mean(), (d > 1e-3).sum())
    assert (dm > golden_stats.POOL_MEAN_TOL).sum() <= 40
