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


def test_softsplat_config5_stress_field(lib):
    """SURVEY 8(d) config 5 — the input bench.py's splat micro-benchmark times: in [1,4,1088,1920] U[0,1), flow i.i.d. N(0, 8 px),
    torch seed 2 (cupy_ops/softsplat.py:140-192's stress case: an INCOHERENT field) — against the plain-C oracle AND, where
    oracle/_ref holds the shape, against an execution of the reference's own kernel text."""
    from oracle import ref_kernels as R

    hp, wp = 1088, 1920
    g = torch.Generator(device="cpu").manual_seed(2)
    x = torch.rand(1, hp, wp, 4, generator=g)                     # exactly bench.py other_paths()'s tensors (NHWC)
    fl = torch.randn(1, hp, wp, 2, generator=g) * 8.0
    a = np.ascontiguousarray(x.numpy().transpose(0, 3, 1, 2))
    f = np.ascontiguousarray(fl.numpy().transpose(0, 3, 1, 2))
    got = _hip_splat(lib, a, f)
    want = _nhwc(M.softsplat_sum(a, f))
    tol = 1e-5 * max(1.0, want.abs().max().item())                # summation order of the scatter differs
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, "config-5 splat vs C oracle")
    # mass: every source lands with bilinear weights summing to 1 unless part of its footprint leaves the frame
    assert got.double().sum().item() <= float(a.astype(np.float64).sum()) * (1 + 1e-6)
    if R.available() and (1, 4, hp, wp) in [tuple(s_) for s_ in R.shapes("softsplat_out")]:
        ref = _nhwc(R.softsplat_out(a, f))
        assert (got - ref).abs().max().item() <= tol, describe_diff(got, ref, "config-5 splat vs the reference kernel text")
        assert (want - ref).abs().max().item() <= tol      # and the C oracle agrees with the text it restates, at this size
    # the far-displacement tail of the same benchmark family: sigma 64 px sends most sources beyond the list kernel's window
    f64 = (f * 8.0).astype(np.float32)
    got = _hip_splat(lib, a, f64)
    want = _nhwc(M.softsplat_sum(a, f64))
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item()), describe_diff(got, want, "sigma-64 splat vs C oracle")


# ---- the list splat's side paths -------------------------------------------------------------------------------------
def _smooth_flow(rng, n, h, w, amp, cells=6):
    import torch.nn.functional as F

    base = torch.from_numpy(rng.standard_normal((n, 2, cells, cells + 2)).astype(np.float32)) * amp
    return F.interpolate(base, size=(h, w), mode="bicubic", align_corners=True).numpy()


@pytest.mark.parametrize("kind", ["smooth", "zoom_in", "to_one_pixel", "rotation", "mixed_far"])
@pytest.mark.parametrize("c", [4, 3, 9])
def test_softsplat_flow_fields(lib, kind, c):
    """coherent fields (the gather's common case), convergent fields (cells with more than SPLAT_K sources spill to the
    global-atomic pass) and far sources, against the sequential oracle"""
    rng = np.random.default_rng(len(kind) + c)
    n, h, w = 2, 96, 160
    a = rng.random((n, c, h, w), dtype=np.float32)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    if kind == "smooth":
        f = _smooth_flow(rng, n, h, w, 6.0)
    elif kind == "zoom_in":            # everything contracts towards the centre by 4x: ~16 sources per target cell
        f = np.stack([np.stack([(w / 2 - xs) * 0.75, (h / 2 - ys) * 0.75])] * n)
    elif kind == "to_one_pixel":       # every source lands in the same 2x2 footprint
        f = np.stack([np.stack([40.3 - xs, 50.6 - ys])] * n)
        f = np.clip(f, -62.0, 62.0)
    elif kind == "rotation":
        th = 0.05
        f = np.stack([np.stack([(np.cos(th) - 1) * (xs - w / 2) - np.sin(th) * (ys - h / 2),
                                np.sin(th) * (xs - w / 2) + (np.cos(th) - 1) * (ys - h / 2)])] * n)
    else:
        f = _smooth_flow(rng, n, h, w, 3.0)
        f[:, :, 10:20, 30:50] += 90.0      # a block of far sources (> 63 px)
        f[0, 0, 5, 5] = np.nan
    f = np.ascontiguousarray(f, np.float32)
    want = _nhwc(M.softsplat_sum(a, f))
    got = _hip_splat(lib, a, f)
    tol = 1e-5 * max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"softsplat {kind} C={c}")


def test_softsplat_is_deterministic_and_exact_for_translations(lib):
    """the gather adds each pixel's contributions in a fixed order (sorted lists): bit-identical run to run; for a uniform
    translation that order is the sequential restatement's (ascending source position) -> bit-exact; for a smooth field the
    orders differ at a few pixels -> last-bit differences only"""
    rng = np.random.default_rng(3)
    a = rng.random((1, 4, 128, 192), dtype=np.float32)
    f = np.ascontiguousarray(_smooth_flow(rng, 1, 128, 192, 4.0), np.float32)
    want = _nhwc(M.softsplat_sum(a, f))
    got, again = _hip_splat(lib, a, f), _hip_splat(lib, a, f)
    assert torch.equal(got, again), "not deterministic run to run"
    assert (got - want).abs().max().item() <= 4 * 2.4e-7 * want.abs().max().item(), describe_diff(got, want, "smooth field: more than 4 ulp")
    f = np.empty((1, 2, 128, 192), np.float32)
    f[:, 0], f[:, 1] = 5.3, -17.6
    want = _nhwc(M.softsplat_sum(a, f))
    got = _hip_splat(lib, a, f)
    assert torch.equal(got, want), describe_diff(got, want, "uniform translation: bit-exact expected")


@pytest.mark.parametrize("opts", ["splat_spill_cap=64", "splat_atomic=1"])
def test_softsplat_fallback_paths_in_a_fresh_process(opts):
    """the spill list overflowing (-> the LDS-atomic tile kernel redoes the launch) and the forced atomic mode (A/B options of
    include/vfi_hip_test.h, applied by the lib fixture from VFI_TEST_OPTIONS — a test-harness variable, not one the library reads):
    the same tests run in a child process"""
    import subprocess
    import sys

    e = dict(os.environ, VFI_TEST_OPTIONS=opts)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k",
                        "flow_fields or vs_c_oracle or prebuilt"], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
