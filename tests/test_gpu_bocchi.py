"""-m gpu: the three node classes on the REAL 1080p demo pair (demo_frames/bocchi0.jpg + bocchi1.jpg of the reference, SURVEY.md
§2 row 25 / §8d config 2(ii)) against what the REAL reference nodes produced for it on torch-CPU (oracle/make_golden_bocchi.py,
oracle/VALIDATION_BOCCHI.log), default synthetic checkpoints and the "hot" ones (flows of 40-100 px, occlusions).

The goldens are fingerprints (oracle/golden_stats.py): per-pixel |d| <= 1e-3 on 12 full-precision 128x128 crops (corners,
borders, centre, interior) and the mean / max of every 8x8 block of the whole frame."""
import json
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
    rounding.  The bound on their number is DERIVED, not chosen: tests/golden/m2m_hot_certificate.json (oracle/m2m_hot_certificate.py)
    records how many values of the ORACLE's own frame move by more than 1e-3 when the flows entering its splats are perturbed by the
    size of this path's measured flow deviation — inside the 12 fingerprint crops: none, for every seed.  So inside the crops the
    plain per-pixel gate holds here too; the full-frame statement is test_m2m_full_frame_vs_host_oracle below."""
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
    cert = json.load(open(os.path.join(golden_dir, "m2m_hot_certificate.json")))[key]
    matched = [v for name, v in cert.items() if name.startswith("flow_rel9e-6")]
    assert (d > 1e-3).sum() <= max(c["crop_values_over_1e-3"] for c in matched), (d.max(), (d > 1e-3).sum())          # = 0: the per-pixel gate
    assert d.mean() <= max(c["crop_mean"] for c in matched) and (dm > golden_stats.POOL_MEAN_TOL).sum() <= max(c["blocks_over_pool_mean_tol"] for c in matched)


# ---- FULL-FRAME per-pixel gates: the oracle (bit-exact with the reference nodes on this very pair, oracle/VALIDATION_BOCCHI.log) is
# executed on the GPU box's HOST cores and every one of the 1080 x 1920 x 3 values is compared — the fingerprints above assert the
# per-pixel gate on 4.7 % of the frame only (VERDICT r3 "weak" 3).


@pytest.mark.parametrize("tag", ["default", "hot"])
def test_rife47_full_frame_vs_host_oracle(hip_lib, frames, tag, oracle_threads):
    from cfi_amd.rife import RifeEngine, run_tasks
    from oracle import rife_oracle

    sd = synth.rife47_synth_state_dict(1234) if tag == "default" else synth.rife47_hot_state_dict(1234)
    want = rife_oracle.rife_vfi(sd, frames, multiplier=4)[1:4]                     # t = .25, .5, .75
    eng = RifeEngine(sd, "4.7")
    try:
        got = run_tasks(eng, frames, [(0, 0.25), (0, 0.5), (0, 0.75)], batch_size=3)
    finally:
        eng.close()
    d = (got - want).abs()
    print(f"RIFE 4.7 {tag} x4 @1080p bocchi, FULL frames vs the oracle on the host: max|d| {d.max().item():.2e} mean {d.mean().item():.2e}")
    assert d.max().item() <= 1e-3


@pytest.mark.parametrize("tag", ["default", "hot"])
def test_film_full_frame_vs_host_oracle(hip_lib, frames, tag, oracle_threads):
    from cfi_amd.film import FilmEngine
    from oracle import film_oracle

    sd = synth.film_synth_state_dict(1234) if tag == "default" else synth.film_hot_state_dict(1234)
    x = frames.permute(0, 3, 1, 2).contiguous()
    with torch.inference_mode():
        want = film_oracle.film_forward(sd, x[0:1], x[1:2]).clamp(0, 1).permute(0, 2, 3, 1)[0]
    eng = FilmEngine(sd)
    try:
        f = frames.cuda()
        got = eng.forward(f[0].contiguous(), f[1].contiguous(), clamp=True).cpu()
    finally:
        eng.close()
    d = (got - want).abs()
    print(f"FILM {tag} x2 @1080p bocchi, FULL frame vs the oracle on the host: max|d| {d.max().item():.2e} mean {d.mean().item():.2e}")
    assert d.max().item() <= 1e-3


@pytest.mark.parametrize("tag,m,k", [("default", 2, 1), ("default", 3, 1), ("hot", 2, 1), ("hot", 3, 1)])
def test_m2m_full_frame_vs_host_oracle(hip_lib, frames, golden_dir, tag, m, k, oracle_threads):
    """Default checkpoint: the plain per-pixel gate on every value of the frame.  Hot checkpoint (refined flows up to 107 px): the
    number of pixels over 1e-3 must not exceed what the ORACLE ITSELF moves when the flows entering its splats are perturbed by the
    size of this path's flow deviation (tests/golden/m2m_hot_certificate.json, 'flow_rel9e-6_*': uniform relative 9e-6 = mean 4.5e-6;
    the smallest count over the seeds is the bound) — and that the deviation of the flows really is of that size is asserted here too."""
    from cfi_amd.m2m import M2MEngine
    from oracle import m2m_model_oracle as MO

    sd = synth.m2m_synth_state_dict(1234) if tag == "default" else synth.m2m_hot_state_dict(1234)
    x = frames.permute(0, 3, 1, 2)
    t = k / m
    with torch.inference_mode():
        (o,), aux = MO.m2m_forward(sd, x[0:1], x[1:2], [torch.full((1, 1, 1, 1), float(t))], return_aux=True)
    want = o[0].permute(1, 2, 0).contiguous()
    eng = M2MEngine(sd)
    try:
        f = frames.cuda()
        eng.prepare(f[0].contiguous(), f[1].contiguous())
        got = eng.render(t).cpu()
        d0, r = eng.d0, eng.r
    finally:
        eng.close()
    d = (got - want).abs()
    over = int((d.max(dim=2).values > 1e-3).sum())
    rel_mean = []
    for di, name in ((0, "ten_fwd"), (1, "ten_bwd")):
        hip = d0[di, :, :, 0:2].repeat(1, 1, 4) + r[di, :, :, 0:8]
        ora = aux[name][0].permute(1, 2, 0)
        rel_mean.append(((hip - ora).abs() / ora.abs().clamp_min(1.0)).mean().item())
    print(f"M2M {tag} x{m} frame {k} @1080p bocchi, FULL frame vs the oracle on the host: max|d| {d.max().item():.2e} mean {d.mean().item():.2e}, "
          f"pixels over 1e-3: {over}; refined flows (max {aux['ten_fwd'].abs().max().item():.0f} px): mean relative deviation {rel_mean[0]:.2e} / {rel_mean[1]:.2e}")
    if tag == "default":
        assert d.max().item() <= 1e-3
        return
    cert = json.load(open(os.path.join(golden_dir, "m2m_hot_certificate.json")))[f"hot_x{m}_{k}"]
    matched = [v for name, v in cert.items() if name.startswith("flow_rel9e-6")]
    assert max(rel_mean) <= 4.5e-6 * 1.25, "the flows deviate by more than the perturbation the certificate was computed for"
    assert over <= min(c["frame_pixels_over_1e-3"] for c in matched), (over, [c["frame_pixels_over_1e-3"] for c in matched])
    assert d.mean().item() <= min(c["frame_mean"] for c in matched)
