"""CPU: the plain-C oracle of M2M's custom ops (oracle/m2m_ops.c, restating the CUDA kernel text) against
  * outputs of the reference's OWN kernels (tests/golden/m2m_ops_ref.npz, written by oracle/validate_m2m_vs_reference.py:
    the reference's cuda_kernel specialises its softsplat_out / costvol_out text, g++ compiles it behind a serial shim);
  * the prebuilt host builds of those kernels (oracle/_ref/*.so via oracle/ref_kernels.py) where present;
  * independent formulations (float64 scatter-add; torch unfold-style shifts)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import m2m_oracle as M


def _splat_ref64(a, f):
    n_, c_, h, w = a.shape
    ref = np.zeros(a.shape, np.float64)
    for n in range(n_):
        for y in range(h):
            for x in range(w):
                X, Y = x + float(f[n, 0, y, x]), y + float(f[n, 1, y, x])
                if not (np.isfinite(X) and np.isfinite(Y)):
                    continue
                x0, y0 = int(np.floor(X)), int(np.floor(Y))
                for xx, yy, wgt in ((x0, y0, (x0 + 1 - X) * (y0 + 1 - Y)), (x0 + 1, y0, (X - x0) * (y0 + 1 - Y)),
                                    (x0, y0 + 1, (x0 + 1 - X) * (Y - y0)), (x0 + 1, y0 + 1, (X - x0) * (Y - y0))):
                    if 0 <= xx < w and 0 <= yy < h:
                        ref[n, :, yy, xx] += a[n, :, y, x] * wgt
    return ref


def test_softsplat_oracle_vs_float64_scatter():
    rng = np.random.default_rng(0)
    a = rng.random((2, 3, 17, 23), dtype=np.float32)
    f = (rng.standard_normal((2, 2, 17, 23)) * 6).astype(np.float32)
    f[0, 0, 3, 4] = np.nan          # skipped (softsplat.py:157-158)
    f[1, 1, 5, 6] = np.inf
    f[0, :, 0, 0] = (-50.0, 3.0)    # lands outside: dropped
    o = M.softsplat_sum(a, f)
    assert np.abs(o - _splat_ref64(a, f)).max() < 2e-5


def test_softsplat_zero_flow_is_identity_and_mass_conserved():
    rng = np.random.default_rng(1)
    a = rng.random((1, 4, 9, 11), dtype=np.float32)
    assert np.array_equal(M.softsplat_sum(a, np.zeros((1, 2, 9, 11), np.float32)), a)
    f = (rng.random((1, 2, 9, 11), dtype=np.float32) - 0.5)  # stays inside for interior pixels
    f[:, :, 0, :] = 0; f[:, :, -1, :] = 0; f[:, :, :, 0] = 0; f[:, :, :, -1] = 0
    o = M.softsplat_sum(a, f)
    assert abs(o.sum() - a.sum()) < 1e-3   # bilinear weights sum to 1


def test_costvol_oracle_vs_shifted_means():
    rng = np.random.default_rng(2)
    one = rng.standard_normal((2, 8, 10, 13)).astype(np.float32)
    two = rng.standard_normal((2, 8, 10, 13)).astype(np.float32)
    cv = torch.from_numpy(M.costvol(one, two))
    t1, t2 = torch.from_numpy(one), torch.from_numpy(two)
    pad = F.pad(t2, (4, 4, 4, 4))
    inb_full = F.pad(torch.ones(1, 1, 10, 13), (4, 4, 4, 4))
    outs = []
    for dy in range(9):
        for dx in range(9):
            sh = pad[:, :, dy:dy + 10, dx:dx + 13]
            inb = inb_full[:, :, dy:dy + 10, dx:dx + 13]
            outs.append((t1 - sh).abs().mean(1, keepdim=True) * inb + t1.abs().mean(1, keepdim=True) * (1 - inb))
    assert (torch.cat(outs, 1) - cv).abs().max().item() < 1e-6
    assert cv.shape == (2, 81, 10, 13)
    # channel 40 = zero displacement
    assert (cv[:, 40] - (t1 - t2).abs().mean(1)).abs().max().item() < 1e-6


# ---- pinned by execution of the reference's kernel text ---------------------------------------------------------------
def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "m2m_ops_ref.npz"))


def test_c_oracle_equals_reference_kernel_outputs(golden_dir):
    g = _golden(golden_dir)
    names = sorted({k[:-4] for k in g.files if k.endswith("_out") and not k.startswith("soft_")})
    assert len([n for n in names if n.startswith("splat")]) >= 7 and len([n for n in names if n.startswith("costvol")]) >= 3
    for name in names:
        a, b, want = g[name + "_a"], g[name + "_b"], g[name + "_out"]
        got = M.softsplat_sum(a, b) if name.startswith("splat") else M.costvol(a, b)
        assert np.array_equal(got, want), f"{name}: oracle/m2m_ops.c differs from the reference kernel's output"


def test_soft_wrapper_equals_reference(golden_dir):
    """softsplat(..., "soft") (cupy_ops/softsplat.py:382-435) as restated in oracle/gmfss_oracle.py"""
    from oracle import gmfss_oracle

    g = _golden(golden_dir)
    got = gmfss_oracle.softsplat_soft(torch.from_numpy(g["soft_in"]), torch.from_numpy(g["soft_flow"]), torch.from_numpy(g["soft_metric"]))
    assert np.array_equal(got.numpy(), g["soft_out"])


def test_c_oracle_equals_prebuilt_reference_kernels():
    """oracle/_ref (built where /root/reference exists, travels with the snapshot): random inputs per manifest shape."""
    from oracle import ref_kernels as R

    if not R.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    rng = np.random.default_rng(5)
    for shp in R.shapes("softsplat_out"):
        if shp[0] * shp[2] * shp[3] > 200000:
            continue
        a = rng.random(shp, dtype=np.float32)
        f = (rng.standard_normal((shp[0], 2, shp[2], shp[3])) * 9).astype(np.float32)
        f[0, 0, 1, 1] = np.nan
        assert np.array_equal(R.softsplat_out(a, f), M.softsplat_sum(a, f)), shp
    for shp in R.shapes("costvol_out"):
        if shp[0] * shp[2] * shp[3] > 5000:
            continue
        one, two = rng.standard_normal(shp).astype(np.float32), rng.standard_normal(shp).astype(np.float32)
        assert np.array_equal(R.costvol_out(one, two), M.costvol(one, two)), shp
