"""CPU: the plain-C oracle of M2M's custom ops (oracle/m2m_ops.c, restating the CUDA kernel text) against
independent formulations (float64 scatter-add; torch unfold-style shifts).  The reference has no CPU path or
tests for these ops, so this is the only executable pin available (parity 'unpinned' per SURVEY 8c)."""
import numpy as np
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
