"""-m gpu: M2M custom ops, HIP (C ABI, NHWC) vs the plain-C oracle (NCHW, restating the CUDA kernel text) AND vs executions of
the reference's own kernel text: tests/golden/m2m_ops_ref.npz and the prebuilt oracle/_ref host kernels (oracle/ref_kernels.py)."""
import os

import numpy as np
import pytest
import torch

from gpu_util import describe_diff, ptr
from oracle import m2m_oracle as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    from cfi_amd import _lib

    _lib.check(hip_lib.vfi_init(0), "vfi_init")
    return hip_lib


def _nhwc(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 3, 1)))


@pytest.mark.parametrize("shape,sigma", [((2, 4, 33, 47), 6.0), ((1, 3, 64, 80), 20.0), ((1, 4, 136, 240), 8.0), ((1, 1, 5, 7), 1.0),
                                         ((1, 12, 70, 90), 3.0), ((1, 4, 200, 300), 60.0)])  # >8 channels: two passes; sigma 60: far-pixel pass
def test_softsplat_vs_c_oracle(lib, shape, sigma):
    from cfi_amd import _lib

    rng = np.random.default_rng(shape[2])
    a = rng.random(shape, dtype=np.float32)
    f = (rng.standard_normal((shape[0], 2, shape[2], shape[3])) * sigma).astype(np.float32)
    if shape[2] > 8:
        f[0, 0, 3, 4] = np.nan
        f[0, 1, 2, 2] = -np.inf
    want = _nhwc(M.softsplat_sum(a, f))
    ad, fd = _nhwc(a).cuda(), _nhwc(f).cuda()
    out = torch.full(want.shape, float("nan"), device="cuda")
    _lib.check(lib.vfi_softsplat_sum(ptr(ad), ptr(fd), ptr(out), shape[0], shape[2], shape[3], shape[1], None), "splat")
    torch.cuda.synchronize()
    got = out.cpu()
    # atomics: summation order differs from the sequential oracle -> tolerance on the accumulated magnitude
    tol = 1e-5 * max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, "softsplat")


@pytest.mark.parametrize("shape", [(2, 32, 17, 30), (1, 32, 34, 60), (1, 8, 5, 7), (1, 20, 40, 37)])
def test_costvol_vs_c_oracle(lib, shape):
    from cfi_amd import _lib

    rng = np.random.default_rng(shape[1] + shape[2])
    one = rng.standard_normal(shape).astype(np.float32)
    two = rng.standard_normal(shape).astype(np.float32)
    want = _nhwc(M.costvol(one, two))
    od, td = _nhwc(one).cuda(), _nhwc(two).cuda()
    n, c, h, w = shape
    # written at channel offset 3 of a wider tensor (the M2M decoder's concat)
    out = torch.full((n, h, w, 90), float("nan"), device="cuda")
    _lib.check(lib.vfi_costvol9x9(ptr(od), c, ptr(td), c, 0, ptr(out), n, h, w, c, 90, 3, None), "costvol")
    torch.cuda.synchronize()
    got = out.cpu()
    assert torch.isnan(got[..., :3]).all() and torch.isnan(got[..., 84:]).all(), "wrote outside its channel window"
    d = (got[..., 3:84] - want).abs().max().item()
    assert d == 0.0, describe_diff(got[..., 3:84], want, "costvol (bit-exact expected)")


# ---- against executions of the reference's own kernels ---------------------------------------------------------------
def _hip_splat(lib, a, f):
    from cfi_amd import _lib

    n, c, h, w = a.shape
    ad, fd = _nhwc(a).cuda(), _nhwc(f).cuda()
    out = torch.full((n, h, w, c), float("nan"), device="cuda")
    _lib.check(lib.vfi_softsplat_sum(ptr(ad), ptr(fd), ptr(out), n, h, w, c, None), "splat")
    torch.cuda.synchronize()
    return out.cpu()


def _hip_costvol(lib, one, two):
    from cfi_amd import _lib

    n, c, h, w = one.shape
    od, td = _nhwc(one).cuda(), _nhwc(two).cuda()
    out = torch.full((n, h, w, 81), float("nan"), device="cuda")
    _lib.check(lib.vfi_costvol9x9(ptr(od), c, ptr(td), c, 0, ptr(out), n, h, w, c, 81, 0, None), "costvol")
    torch.cuda.synchronize()
    return out.cpu()


def test_hip_ops_vs_reference_kernel_goldens(lib, golden_dir):
    g = np.load(os.path.join(golden_dir, "m2m_ops_ref.npz"))
    names = sorted({k[:-4] for k in g.files if k.endswith("_out") and not k.startswith("soft_")})
    for name in names:
        a, b, want = g[name + "_a"], g[name + "_b"], _nhwc(g[name + "_out"])
        if name.startswith("splat"):
            got = _hip_splat(lib, a, b)
            tol = 1e-5 * max(1.0, want.abs().max().item())       # summation order of the scatter differs
            assert (got - want).abs().max().item() <= tol, describe_diff(got, want, name)
        else:
            got = _hip_costvol(lib, a, b)
            assert torch.equal(got, want), describe_diff(got, want, name + " (bit-exact expected)")


def test_hip_ops_vs_prebuilt_reference_kernels(lib):
    from oracle import ref_kernels as R

    if not R.available():
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    rng = np.random.default_rng(11)
    for shp in R.shapes("softsplat_out"):
        for sigma in (2.0, 12.0):
            a = rng.random(shp, dtype=np.float32)
            f = (rng.standard_normal((shp[0], 2, shp[2], shp[3])) * sigma).astype(np.float32)
            f[0, 1, 2, 3] = np.inf
            want = _nhwc(R.softsplat_out(a, f))
            got = _hip_splat(lib, a, f)
            tol = 1e-5 * max(1.0, want.abs().max().item())
            assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"softsplat {shp} sigma {sigma}")
    for shp in R.shapes("costvol_out"):
        one, two = rng.standard_normal(shp).astype(np.float32), rng.standard_normal(shp).astype(np.float32)
        want = _nhwc(R.costvol_out(one, two))
        got = _hip_costvol(lib, one, two)
        assert torch.equal(got, want), describe_diff(got, want, f"costvol {shp} (bit-exact expected)")
