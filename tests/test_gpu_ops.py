"""-m gpu: single-kernel parity, HIP (through the C ABI) vs the oracle's torch-CPU ops.

Tolerances: the warp reproduces the reference's fp32 expression order (<= 2e-6 abs on [0,1]
data); convolutions differ from oneDNN only in fp32 summation order (<= 2e-5 relative to the
output scale).  Both are far inside the path's |d| <= 1e-3 contract."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import describe_diff, hptr, nchw, nhwc, ptr
from oracle import rife_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    from cfi_amd import _lib

    assert torch.cuda.is_available()
    _lib.check(hip_lib.vfi_init(0), "vfi_init")
    return hip_lib


def _check(lib, rc, what):
    from cfi_amd import _lib

    _lib.check(rc, what)


@pytest.mark.parametrize("shape,mag", [((2, 4, 40, 56), 30.0), ((1, 3, 128, 192), 6.0), ((1, 8, 67, 93), 200.0)])
def test_warp_vs_oracle(lib, shape, mag):
    n, c, h, w = shape
    g = torch.Generator().manual_seed(h * w)
    x = torch.rand(n, c, h, w, generator=g)
    fl = (torch.rand(n, 2, h, w, generator=g) - 0.5) * mag
    want = nhwc(rife_oracle.warp(x, fl))
    xd, fd = nhwc(x).cuda(), nhwc(fl).cuda()
    out = torch.empty_like(xd)
    _check(lib, lib.vfi_warp_border(ptr(xd), ptr(fd), ptr(out), n, h, w, c, None), "vfi_warp_border")
    torch.cuda.synchronize()
    got = out.cpu()
    assert (got - want).abs().max().item() <= 2e-6, describe_diff(got, want, "warp")


def test_warp_vs_reference_golden(lib, golden_dir):
    import os

    g = np.load(os.path.join(golden_dir, "rife_warp.npz"))
    x, fl, y = torch.from_numpy(g["x"]), torch.from_numpy(g["flow"]), torch.from_numpy(g["y"])
    n, c, h, w = x.shape
    xd, fd = nhwc(x).cuda(), nhwc(fl).cuda()
    out = torch.empty_like(xd)
    _check(lib, lib.vfi_warp_border(ptr(xd), ptr(fd), ptr(out), n, h, w, c, None), "vfi_warp_border")
    got = out.cpu()
    assert (got - nhwc(y)).abs().max().item() <= 2e-6, describe_diff(got, nhwc(y), "warp golden")


def _conv_case(lib, n, h, w, cin, cout, stride, res, variant, seed=0, naive=False):
    g = torch.Generator().manual_seed(seed + cin * 131 + cout)
    x = torch.rand(n, cin, h, w, generator=g) * 2 - 1
    wt = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    b = torch.rand(cout, generator=g) - 0.5
    beta = (0.5 + torch.rand(cout, generator=g)) if res else None
    y = F.conv2d(x, wt, b, stride, 1)
    if res:
        y = y * beta.view(1, -1, 1, 1) + x
    want = nhwc(F.leaky_relu(y, 0.2))
    xd = nhwc(x).cuda()
    out = torch.full(want.shape, float("nan"), device="cuda")
    if naive:
        rc = lib.vfi_conv3x3_naive(ptr(xd), hptr(wt), hptr(b), hptr(beta), ptr(out), n, h, w, cin, cout, stride, 1, 0.2, None)
    else:
        rc = lib.vfi_conv3x3(ptr(xd), hptr(wt), hptr(b), hptr(beta), ptr(out), n, h, w, cin, cout, stride, 1, 0.2, variant, None)
    _check(lib, rc, "vfi_conv3x3")
    torch.cuda.synchronize()
    got = out.cpu()
    tol = 2e-5 * max(1.0, want.abs().max().item())
    assert not torch.isnan(got).any(), "NaN / unwritten outputs: " + describe_diff(torch.nan_to_num(got, nan=1e9), want, "conv")
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"conv cin={cin} cout={cout} s={stride} v={variant}")


# (variant, cin, cout): every tile configuration of csrc/conv_mfma.hip at least once
S1 = [(0, 64, 64), (1, 96, 96), (2, 64, 128), (3, 96, 192), (4, 32, 32), (5, 16, 32), (6, 64, 64), (7, 128, 128),
      # second generation (LDS-DMA, csrc/conv_mfma2.hip)
      (32, 64, 64), (33, 96, 96), (34, 64, 128), (35, 96, 192), (36, 128, 128), (37, 128, 128), (38, 64, 64), (32, 8, 64)]
S2 = [(8, 24, 64), (9, 16, 96), (10, 24, 32), (11, 32, 64), (8, 24, 48),
      (39, 24, 64), (40, 32, 64), (41, 16, 96), (42, 24, 32), (39, 24, 48)]


@pytest.mark.parametrize("variant,cin,cout", S1)
def test_conv3x3_s1_variants(lib, variant, cin, cout):
    # 37x45: ragged against every tile shape; residual path only where Cin == Cout
    _conv_case(lib, 2, 37, 45, cin, cout, 1, cin == cout, variant)


@pytest.mark.parametrize("variant,cin,cout", S2)
def test_conv3x3_s2_variants(lib, variant, cin, cout):
    _conv_case(lib, 2, 38, 50, cin, cout, 2, False, variant)


@pytest.mark.parametrize("cin,cout,stride", [(64, 64, 1), (192, 192, 1), (20, 32, 2), (48, 96, 2)])
def test_conv3x3_heuristic_and_naive(lib, cin, cout, stride):
    _conv_case(lib, 1, 34, 60, cin, cout, stride, stride == 1, -1)
    _conv_case(lib, 1, 34, 60, cin, cout, stride, stride == 1, -1, naive=True)


def test_conv3x3_tiny_and_exact_tile(lib):
    _conv_case(lib, 1, 4, 8, 16, 32, 1, False, 4)      # one sub-tile exactly
    _conv_case(lib, 1, 1, 1, 16, 32, 1, False, 4)      # single pixel
    _conv_case(lib, 3, 16, 16, 64, 64, 1, True, 0)     # exactly one 16x16 tile per image


# ---- Winograd F(2x2,3x3) form of the 3x3 stride-1 convolution (csrc/conv_wino.hip): variant 100 = 16x8-pixel regions per wave,
# 101 = 32x4.  Tolerance as for the direct kernels: fp32 round-off only (the transforms add / subtract and halve, nothing else).
WINO = [(100, 64, 64, 2, 37, 45, True), (100, 8, 32, 1, 8, 16, False), (100, 24, 40, 3, 19, 33, False), (100, 192, 192, 1, 34, 60, True),
        (101, 64, 64, 2, 37, 45, True), (101, 96, 96, 1, 34, 60, True), (101, 16, 128, 1, 5, 70, False), (100, 64, 64, 1, 1, 1, False),
        (100, 128, 128, 2, 68, 120, True), (101, 128, 128, 2, 68, 120, False), (100, 64, 64, 9, 272, 480, False)]


@pytest.mark.parametrize("variant,cin,cout,n,h,w,res", WINO)
def test_conv3x3_winograd(lib, variant, cin, cout, n, h, w, res):
    _conv_case(lib, n, h, w, cin, cout, 1, res, variant)


@pytest.mark.parametrize("act,post,pad_mode,res", [(0, None, 0, False), (1, None, 0, False), (3, None, 1, False), (4, (0.8, 0.1), 0, False), (5, None, 0, False),
                                                   (3, None, 0, True), (1, (2.0, -0.5), 1, True), (2, None, 0, False)])
def test_layer_object_winograd_matches_direct_and_torch(lib, act, post, pad_mode, res):
    """One vfi_conv_create_ex layer (3x3, stride 1; chan_map, channel windows, zero / replicate padding, every activation of the
    generic epilogue, residual-before-activation, post affine) run in BOTH forms on one input (vfi_test_conv_algo) vs torch."""
    import ctypes as C

    g = torch.Generator().manual_seed(act * 10 + pad_mode)
    n, h, w, cin, cout = 2, 27, 41, 20, 48
    x = torch.rand(n, cin, h, w, generator=g) * 2 - 1
    wt = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) / (cin * 9) ** 0.5
    b = torch.rand(cout, generator=g) - 0.5
    slopes = -0.5 + 2.0 * torch.rand(cout, generator=g)        # any sign / size: r5's per-channel hot epilogue (conv_wino MODE 1) is an exact PReLU
    r = torch.rand(n, cout, h, w, generator=g) - 0.5
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate" if pad_mode else "constant")
    y = F.conv2d(xp.double(), wt.double(), b.double())
    if res:
        y = y + r.double()
    if act == 1:
        y = F.leaky_relu(y, 0.2)
    elif act == 2:
        y = y.clamp(0, 1)
    elif act == 3:
        y = torch.where(y > 0, y, y * slopes.double().view(1, -1, 1, 1))
    elif act == 4:
        y = torch.sigmoid(y)
    elif act == 5:
        y = F.gelu(y)
    if post:
        y = y * post[0] + post[1]
    want = nhwc(y.float())
    cphys = 24
    cmap = [cphys - 1 - c for c in range(cin)]
    xin = torch.rand(n, h, w, cphys + 8, generator=g)           # garbage in the unmapped / outer channels
    xin[..., 4:4 + cphys][..., cmap] = x.permute(0, 2, 3, 1)
    xin[..., 4:4 + cphys][..., [c for c in range(cphys) if c not in cmap]] = 0.0
    xd, rd = xin.cuda(), nhwc(r).cuda().contiguous()
    cm = (C.c_int * cin)(*cmap)
    hnd = lib.vfi_conv_create_ex(0, wt.data_ptr(), b.data_ptr(), cout, cin, 3, 1, pad_mode, cm, cphys, slopes.data_ptr() if act == 3 else None)
    assert hnd, "create failed"
    outs = {}
    try:
        for mode in (1, 2):
            assert lib.vfi_test_conv_algo(mode) == mode
            out = torch.full((n, h, w, cout + 5), float("nan"), device="cuda")
            _check(lib, lib.vfi_conv_forward_ex(hnd, xd.data_ptr() + 16, cphys + 8, h, w, out.data_ptr() + 12, cout + 5, n, act, 0.2,
                                                post[0] if post else 0.0, post[1] if post else 0.0, rd.data_ptr() if res else None, cout, None), "conv_forward_ex")
            torch.cuda.synchronize()
            got = out.cpu()
            assert torch.isnan(got[..., :3]).all() and torch.isnan(got[..., 3 + cout:]).all(), "wrote outside its channel window"
            outs[mode] = got[..., 3:3 + cout]
    finally:
        lib.vfi_test_conv_algo(0)
        lib.vfi_conv_destroy(hnd)
    tol = 2e-5 * max(1.0, want.abs().max().item())
    for mode, got in outs.items():
        assert not torch.isnan(got).any(), f"mode {mode}: unwritten outputs"
        assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"mode {mode} act {act}")


@pytest.mark.parametrize("act", [0, 1, 3])
@pytest.mark.parametrize("cin,cout,n,h,w", [(64, 16, 2, 40, 56), (24, 32, 1, 33, 70), (128, 32, 2, 17, 30), (256, 64, 1, 34, 60),
                                            (64, 16, 1, 136, 240), (24, 32, 1, 131, 133)])   # r5: the last two are >= 128 x 128 input pixels: the Winograd form's sizes
def test_layer_object_deconv_winograd_matches_direct_and_torch(lib, act, cin, cout, n, h, w):
    """vfi_conv_create_ex(kind = 1): ConvTranspose2d(4, 2, 1) of the layer objects (M2M / IFRNet / IFUNet / GMFSS decoders) in both forms —
    the grouped direct kernel (A/B option deconv_wino = 0) and ONE 3x3 layer with 4 * Cout channels on the Winograd kernel whose
    epilogue interleaves the parities into NHWC at twice the resolution (default from 128 x 128 input pixels up) — vs torch: none / LeakyReLU /
    per-channel PReLU with slopes of any sign and size, a channel window on both sides, odd sizes (partial regions), small layers that
    stay on the direct kernel by size.  The two forms must also DIFFER where the Winograd form is expected to run (the A/B option is live)."""
    g = torch.Generator().manual_seed(cin + cout + act)
    x = torch.rand(n, cin, h, w, generator=g) * 2 - 1
    wt = ((torch.rand(cin, cout, 4, 4, generator=g) * 2 - 1) / (cin * 4) ** 0.5).contiguous()
    b = (torch.rand(cout, generator=g) - 0.5).contiguous()
    slopes = (-0.5 + 2.0 * torch.rand(cout, generator=g)).contiguous()
    y = F.conv_transpose2d(x.double(), wt.double(), b.double(), 2, 1)
    if act == 1:
        y = F.leaky_relu(y, 0.25)
    elif act == 3:
        y = torch.where(y > 0, y, y * slopes.double().view(1, -1, 1, 1))
    want = nhwc(y.float())
    xin = torch.rand(n, h, w, cin + 8, generator=g)
    xin[..., 4:4 + cin] = x.permute(0, 2, 3, 1)
    xd = xin.cuda()
    hnd = lib.vfi_conv_create_ex(1, wt.data_ptr(), b.data_ptr(), cout, cin, 4, 2, 0, None, cin, slopes.data_ptr() if act == 3 else None)
    assert hnd, "create failed"
    outs = {}
    try:
        for form in (1, 0):
            assert lib.vfi_test_set_option(b"deconv_wino", form) == 0
            out = torch.full((n, 2 * h, 2 * w, cout + 5), float("nan"), device="cuda")
            _check(lib, lib.vfi_conv_forward_ex(hnd, xd.data_ptr() + 16, cin + 8, h, w, out.data_ptr() + 12, cout + 5, n, act, 0.25, 0.0, 0.0, None, 0, None),
                   "conv_forward_ex")
            torch.cuda.synchronize()
            got = out.cpu()
            assert torch.isnan(got[..., :3]).all() and torch.isnan(got[..., 3 + cout:]).all(), "wrote outside its channel window"
            outs[form] = got[..., 3:3 + cout]
    finally:
        lib.vfi_test_set_option(b"deconv_wino", 1)
        lib.vfi_conv_destroy(hnd)
    tol = 2e-5 * max(1.0, want.abs().max().item())
    for form, got in outs.items():
        assert not torch.isnan(got).any(), f"deconv_wino={form}: unwritten outputs"
        assert (got - want).abs().max().item() <= tol, describe_diff(got, want, f"deconv_wino={form} act {act}")
    if h * w >= 16384:      # the 3x3 / Winograd form really ran: its summation order differs from the grouped direct kernel's
        assert not torch.equal(outs[0], outs[1]), "deconv_wino = 1 took the direct kernel at a size where the Winograd form is the rule"
    else:
        assert torch.equal(outs[0], outs[1]), "below 128 x 128 input pixels both settings take the direct kernel"


@pytest.mark.parametrize("gvariant", [None, "direct", 12, 13, 43, 44])
@pytest.mark.parametrize("cin,h,w", [(64, 17, 30), (192, 9, 15), (96, 34, 60)])
def test_deconv4x4_pixelshuffle(lib, cin, h, w, gvariant):
    # None: the default form — the layer as a 96-channel 3x3 convolution on the Winograd kernel with the pixel-shuffle epilogue
    # (A/B option deconv_wino = 1); "direct": the grouped direct kernel (deconv_wino = 0) with its own tile choice; a number: that
    # tile variant of the grouped kernel forced (A/B option grouped_variant, read at every launch)
    try:
        if gvariant == "direct":
            assert lib.vfi_test_set_option(b"deconv_wino", 0) == 0
        elif gvariant is not None:
            assert lib.vfi_test_set_option(b"grouped_variant", gvariant) == 0
        _deconv_case(lib, cin, h, w)
    finally:
        lib.vfi_test_set_option(b"grouped_variant", -1)
        lib.vfi_test_set_option(b"deconv_wino", 1)


def _deconv_case(lib, cin, h, w):
    g = torch.Generator().manual_seed(cin)
    x = torch.rand(2, cin, h, w, generator=g) * 2 - 1
    wt = (torch.rand(cin, 24, 4, 4, generator=g) * 2 - 1) / (cin * 4) ** 0.5
    b = torch.rand(24, generator=g) - 0.5
    want = nhwc(F.pixel_shuffle(F.conv_transpose2d(x, wt, b, 2, 1), 2))
    xd = nhwc(x).cuda()
    out = torch.full(want.shape, float("nan"), device="cuda")
    _check(lib, lib.vfi_deconv4x4_ps2(ptr(xd), hptr(wt), hptr(b), ptr(out), 2, h, w, cin, 24, None), "vfi_deconv4x4_ps2")
    got = out.cpu()
    assert not torch.isnan(got).any()
    tol = 2e-5 * max(1.0, want.abs().max().item())
    assert (got - want).abs().max().item() <= tol, describe_diff(got, want, "deconv+ps")


def test_clock_probe_records(lib):
    """vfi_clock_probe (include/vfi_hip.h, r5): while a buffer is installed every launch of the Winograd kernel fills one record —
    s_memtime / s_memrealtime of workgroup 0 at its start and end + the host's launch index — and bench.py turns the two deltas into the
    shader clock.  Checks the record contents, the names, that a full buffer stops recording, and that uninstalling stops it."""
    import ctypes as C

    from cfi_amd import _lib

    n, h, w, c = 4, 136, 240, 64
    x = torch.rand(n, h, w, c, device="cuda")
    wt, b = (torch.rand(c, c, 3, 3) - 0.5) * 0.1, torch.rand(c) - 0.5
    out = torch.empty(n, h, w, c, device="cuda")

    def launch():
        _check(lib, lib.vfi_conv3x3(C.c_void_p(x.data_ptr()), C.c_void_p(wt.data_ptr()), C.c_void_p(b.data_ptr()), None, C.c_void_p(out.data_ptr()),
                                    n, h, w, c, c, 1, 1, 0.2, 100, None), "vfi_conv3x3")

    launch()
    torch.cuda.synchronize()
    rec = torch.zeros((3, 8), dtype=torch.int64, device="cuda")
    _check(lib, lib.vfi_clock_probe(C.c_void_p(rec.data_ptr()), 3), "vfi_clock_probe")
    try:
        for _ in range(5):          # two more launches than records: the extra ones must not write anywhere
            launch()
        torch.cuda.synchronize()
    finally:
        _check(lib, lib.vfi_clock_probe(None, 0), "vfi_clock_probe off")
    names = _lib.clock_probe_names()
    assert len(names) == 3 and len(set(names)) == 1, names      # r6: a full buffer stops the names too
    r = rec.cpu().numpy().astype("uint64")
    for i, (t0, r0, t1, r1, tag, *rest) in enumerate(r.tolist()):
        assert t1 > t0 and r1 > r0 and tag == i and rest == [0, 0, 0], (i, t0, r0, t1, r1, tag, rest)
        mhz = (t1 - t0) / (r1 - r0) * 100.0
        assert 400.0 < mhz < 2600.0, mhz          # s_memrealtime = 100 MHz; the shader clock of an MI355X under load
    before = rec.clone()
    launch()
    torch.cuda.synchronize()
    assert torch.equal(rec, before), "a launch after the probe was uninstalled wrote a record"


def test_wino_probe_forms_are_bit_identical(lib):
    """The cycle-ledger forms of the hot Winograd instantiation (test option wino_probe = 1..4, tools/wino_ledger.py,
    docs/design/winograd.md section 5b) compute exactly what the product kernel computes, and each leaves its stamp sums."""
    import ctypes as C

    n, h, w, c = 8, 136, 240, 64
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(n, h, w, c, generator=g) - 0.5).cuda()
    wt, b = (torch.rand(c, c, 3, 3, generator=g) - 0.5) * 0.1, torch.rand(c, generator=g) - 0.5

    def run(probe):
        assert lib.vfi_test_set_option(b"wino_probe", probe) == 0
        out = torch.empty(n, h, w, c, device="cuda")
        _check(lib, lib.vfi_conv3x3(C.c_void_p(x.data_ptr()), C.c_void_p(wt.data_ptr()), C.c_void_p(b.data_ptr()), None, C.c_void_p(out.data_ptr()),
                                    n, h, w, c, c, 1, 1, 0.2, 100, None), "vfi_conv3x3")
        torch.cuda.synchronize()
        return out

    try:
        base = run(0)
        for probe in (1, 2, 3, 4):
            out = run(probe)
            assert torch.equal(out, base), f"probe form {probe} changed the output"
            sums = (C.c_uint32 * 32)()
            _check(lib, lib.vfi_test_wino_probe_read(sums), "vfi_test_wino_probe_read")
            for wave in range(4):
                q = list(sums[wave * 8:(wave + 1) * 8])
                assert q[7] == probe and q[4] > 0 and q[6] > 0, (probe, wave, q)      # probe id, stamps taken, chunks walked
    finally:
        lib.vfi_test_set_option(b"wino_probe", 0)
