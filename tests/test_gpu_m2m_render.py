"""-m gpu: M2M's one-kernel render (csrc/m2m_render.hip: vfi_m2m_photo_tiles + vfi_m2m_render_fused) against
  * the three-step form it replaces (vfi_m2m_splat_inputs -> vfi_softsplat_sum x 8 -> vfi_m2m_combine): BIT-IDENTICAL on coherent
    fields (every tile's source window fits one LDS stage), within summation-order noise beyond;
  * the oracle: forwarp_mframe_mask restated with the plain-C splat of oracle/m2m_ops.c (M2M_arch.py:551-581, :1012-1037;
    cupy_ops/softsplat.py:140-192) on crafted fields — far displacements (the old path's far pass), convergent fields (cells with more
    than 6 sources: the scan path), i.i.d. noise (windows of several strips), non-finite flows;
and the tiled photometric kernel against vfi_m2m_photo (bit-identical tf / e) and numpy tile ranges.  Through the C ABI only."""
import numpy as np
import pytest
import torch

from gpu_util import describe_diff, ptr
from cfi_amd import synth
from oracle import m2m_oracle as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib(hip_lib):
    from cfi_amd import _lib

    _lib.check(hip_lib.vfi_init(0), "vfi_init")
    return hip_lib


def _ck(rc, what):
    from cfi_amd import _lib

    _lib.check(rc, what)


def tile_ranges(tf):
    """tf [8,H,W,2] (numpy) -> ([8,tiles,4], [8]) as vfi_m2m_photo_tiles defines them."""
    _, H, W, _ = tf.shape
    ty, tx = -(-H // 32), -(-W // 32)
    out = np.empty((8, ty * tx, 4), np.float32)
    smax = np.zeros(8, np.float32)
    for s in range(8):
        for j in range(ty):
            for i in range(tx):
                blk = tf[s, 32 * j:32 * j + 32, 32 * i:32 * i + 32].reshape(-1, 2)
                ok = np.isfinite(blk).all(axis=1)
                if ok.any():
                    b = blk[ok]
                    out[s, j * tx + i] = (b[:, 0].min(), b[:, 0].max(), b[:, 1].min(), b[:, 1].max())
                    smax[s] = max(smax[s], np.abs(b).max())
                else:
                    out[s, j * tx + i] = (3e38, -3e38, 3e38, -3e38)
    return out, smax


def three_step(lib, d0, tf, e, stats, t, H, W):
    """The form round 5 shipped, through the same C entry points."""
    _, Hp, Wp, _ = d0.shape
    sin = torch.empty((8, Hp, Wp, 4), device="cuda")
    sfl = torch.empty((8, Hp, Wp, 2), device="cuda")
    sout = torch.empty((8, Hp, Wp, 4), device="cuda")
    out = torch.full((H, W, 3), float("nan"), device="cuda")
    _ck(lib.vfi_m2m_splat_inputs(ptr(d0), 8, ptr(tf), ptr(e), t, ptr(sin), ptr(sfl), Hp, Wp, None), "splat_inputs")
    _ck(lib.vfi_softsplat_sum(ptr(sin), ptr(sfl), ptr(sout), 8, Hp, Wp, 4, None), "softsplat")
    _ck(lib.vfi_m2m_combine(ptr(sout), ptr(d0), 8, ptr(stats), t, ptr(out), Hp, Wp, H, W, None), "combine")
    torch.cuda.synchronize()
    return out.cpu()


def fused(lib, d0, tf, e, stats, t, H, W, ranges=None, smax=None):
    _, Hp, Wp, _ = d0.shape
    img4 = torch.cat([d0[..., 2:5], torch.ones_like(d0[..., :1])], -1).contiguous()
    if ranges is None:
        r_, s_ = tile_ranges(tf.cpu().numpy())
        ranges, smax = torch.from_numpy(r_).cuda(), torch.from_numpy(s_).cuda()
    out = torch.full((H, W, 3), float("nan"), device="cuda")
    _ck(lib.vfi_m2m_render_fused(ptr(img4), ptr(tf), ptr(e), ptr(ranges), ptr(smax), ptr(stats), t, ptr(out), Hp, Wp, H, W, None), "render_fused")
    torch.cuda.synchronize()
    return out.cpu()


def oracle_render(d0, tf, e, stats, t, H, W):
    """forwarp_mframe_mask + hole fill + de-normalisation with the plain-C splat (M2M_arch.py:551-581, :1012-1037), fp32 like the reference."""
    d0, tf, e = d0.cpu().numpy(), tf.cpu().numpy(), e.cpu().numpy()
    mean, sd = (float(v) for v in stats.cpu())
    _, Hp, Wp, _ = d0.shape
    t32 = np.float32(t)
    t1 = np.float32(1.0) - t32
    acc = np.zeros((3, Hp, Wp), np.float32)
    norm = np.zeros((Hp, Wp), np.float32)
    for b in range(4):
        o = []
        for d in range(2):
            s = 2 * b + d
            td, tm = (t1, t32) if d == 0 else (t32, t1)
            img = d0[d, :, :, 2:5].transpose(2, 0, 1)
            inp = np.concatenate([(img * td) * e[s][None], (td * e[s])[None]], 0)[None].astype(np.float32)
            fl = (tf[s] * tm).transpose(2, 0, 1)[None].astype(np.float32)
            o.append(M.softsplat_sum(np.ascontiguousarray(inp), np.ascontiguousarray(fl))[0])
        acc += o[0][:3] + o[1][:3]
        norm += (o[0][3] + np.float32(1e-7)) + (o[1][3] + np.float32(1e-7))
    with np.errstate(divide="ignore", invalid="ignore"):
        v = acc / norm[None]
    hole = norm < np.float32(0.00001)
    fill = t1 * d0[0, :, :, 2:5].transpose(2, 0, 1) + t32 * d0[1, :, :, 2:5].transpose(2, 0, 1)
    v = np.where(hole[None], v + fill, v)
    out = v * np.float32(sd) + np.float32(mean)
    return torch.from_numpy(np.ascontiguousarray(out[:, :H, :W].transpose(1, 2, 0)))


def make_case(Hp, Wp, kind, seed):
    g = torch.Generator().manual_seed(seed)
    d0 = torch.zeros((2, Hp, Wp, 8))
    d0[..., 2:5] = torch.randn((2, Hp, Wp, 3), generator=g)
    yy, xx = torch.meshgrid(torch.arange(Hp, dtype=torch.float32), torch.arange(Wp, dtype=torch.float32), indexing="ij")
    tf = torch.empty((8, Hp, Wp, 2))
    for s in range(8):
        if kind == "smooth":      # coherent: a slow wave of a few pixels on a constant offset, different per splat
            tf[s, ..., 0] = 3.0 * torch.sin(yy / 47.0 + s) + 2.5 * torch.cos(xx / 53.0) + (s - 3.5) * 1.7
            tf[s, ..., 1] = 2.0 * torch.cos(yy / 43.0) - 3.0 * torch.sin(xx / 59.0 + 0.5 * s) - (s - 3.5) * 0.9
        elif kind == "wavy":      # still coherent, but a tile's flow range of ~8 px makes windows of more than one strip
            tf[s, ..., 0] = 5.0 * torch.sin(yy / 11.0 + s) + 4.5 * torch.cos(xx / 13.0) + (s - 3.5) * 1.7
            tf[s, ..., 1] = 4.0 * torch.cos(yy / 9.0) - 5.0 * torch.sin(xx / 12.0 + 0.5 * s) - (s - 3.5) * 0.9
        elif kind == "translate":
            tf[s, ..., 0], tf[s, ..., 1] = 2.25 * (s + 1), -1.75 * (s + 1)
        elif kind == "far":       # displacements up to ~150 px (the three-step form's far pass), still coherent
            tf[s, ..., 0] = 150.0 * torch.sin(yy / 90.0 + s) + 0.3 * torch.cos(xx / 7.0)
            tf[s, ..., 1] = -110.0 * torch.cos(xx / 120.0 + 0.3 * s)
        elif kind == "noise":     # incoherent, sigma 8 px: windows of several strips
            tf[s] = torch.randn((Hp, Wp, 2), generator=g) * 8.0
        elif kind == "zoom":      # strongly convergent: 16+ sources per target cell
            tf[s, ..., 0] = (Wp / 2 - xx) * 0.78 + 0.1 * s
            tf[s, ..., 1] = (Hp / 2 - yy) * 0.78 - 0.1 * s
        elif kind == "point":     # everything lands in one pixel's footprint
            tf[s, ..., 0] = (Wp / 3 + 0.37 * (s + 1)) - xx
            tf[s, ..., 1] = (Hp / 2 + 0.21 * (s + 1)) - yy
        else:
            raise ValueError(kind)
    e = torch.exp(torch.rand((8, Hp, Wp), generator=g) * 2.0 - 1.0)
    stats = torch.tensor([0.45, 0.27])
    return d0.cuda(), tf.cuda(), e.cuda(), stats.cuda()


@pytest.mark.parametrize("Hp,Wp,H,W,kind,t", [(64, 128, 60, 121, "smooth", 0.5), (128, 192, 128, 192, "smooth", 0.25), (64, 64, 64, 64, "translate", 0.5),
                                              (192, 256, 180, 250, "smooth", 2.0 / 3.0), (64, 128, 64, 128, "smooth", 0.0), (64, 128, 64, 128, "smooth", 1.0)])
def test_fused_render_is_bit_identical_to_three_step_on_coherent_fields(lib, Hp, Wp, H, W, kind, t):
    d0, tf, e, stats = make_case(Hp, Wp, kind, seed=Hp + Wp)
    want = three_step(lib, d0, tf, e, stats, t, H, W)
    got = fused(lib, d0, tf, e, stats, t, H, W)
    assert torch.equal(got, want), describe_diff(got, want, f"fused vs three-step {kind} t={t}")
    if kind == "translate":      # the sequential oracle's order for a uniform translation: bit-exact too
        ora = oracle_render(d0, tf, e, stats, t, H, W)
        assert torch.equal(got, ora), describe_diff(got, ora, "fused vs oracle, uniform translation")


@pytest.mark.parametrize("Hp,Wp,kind,t", [(192, 320, "far", 0.5), (192, 320, "far", 0.9), (128, 192, "noise", 0.5), (128, 128, "zoom", 0.5), (128, 128, "zoom", 0.9),
                                          (96, 160, "point", 0.5), (128, 192, "smooth", 0.4), (128, 192, "wavy", 0.3)])
def test_fused_render_vs_oracle_on_hard_fields(lib, Hp, Wp, kind, t):
    d0, tf, e, stats = make_case(Hp, Wp, kind, seed=7 + Hp)
    want = oracle_render(d0, tf, e, stats, t, Hp, Wp)
    got = fused(lib, d0, tf, e, stats, t, Hp, Wp)
    old = three_step(lib, d0, tf, e, stats, t, Hp, Wp)
    # pixels whose whole weight is below the hole threshold flip on the last bit of a sum: compare where the normaliser is well away from it
    fin = torch.isfinite(want) & torch.isfinite(got)
    scale = max(1.0, float(want[fin].abs().max()))
    d = (got - want).abs()[fin]
    d_old = (old - want).abs()[fin]
    frac_bad = float((d > 2e-4 * scale).float().mean())
    assert frac_bad <= 2e-4 and float(d.mean()) <= 1e-6 * scale, (describe_diff(got, want, f"fused vs oracle {kind}"), f"three-step: max {float(d_old.max()):.2e}")
    assert bool((torch.isfinite(want) == torch.isfinite(got)).all())


def test_fused_render_nonfinite_flows_never_splat(lib):
    d0, tf, e, stats = make_case(64, 128, "smooth", seed=3)
    tf[0, 5, 7, 0] = float("nan")
    tf[3, 20, 100, 1] = float("inf")
    tf[6, 40:44, 60:64, :] = float("-inf")
    tf[7, 0:32, 0:32, :] = float("nan")      # a whole tile without a finite flow: its range entry is inverted
    want = three_step(lib, d0, tf, e, stats, 0.5, 64, 128)
    got = fused(lib, d0, tf, e, stats, 0.5, 64, 128)
    assert torch.equal(got, want), describe_diff(got, want, "non-finite flows")


def test_photo_tiles_matches_photo_and_numpy_ranges(lib):
    g = torch.Generator().manual_seed(5)
    Hp, Wp = 128, 192
    d0 = torch.randn((2, Hp, Wp, 8), generator=g)
    d0[..., 0:2] *= 6.0
    r = torch.randn((2, Hp, Wp, 12), generator=g)
    r[..., 0:8] *= 2.0
    d0d, rd = d0.cuda(), r.cuda()
    tf0, e0 = torch.empty((8, Hp, Wp, 2), device="cuda"), torch.empty((8, Hp, Wp), device="cuda")
    _ck(lib.vfi_m2m_photo(ptr(d0d), 8, ptr(rd), 12, 0.7, ptr(tf0), ptr(e0), Hp, Wp, None), "photo")
    tf1, e1 = torch.empty_like(tf0), torch.empty_like(e0)
    img4 = torch.empty((2, Hp, Wp, 4), device="cuda")
    _ck(lib.vfi_m2m_image4(ptr(d0d), 8, ptr(img4), Hp, Wp, None), "image4")
    # the image warp through the compact plane == vfi_warp_m2m on d0's image channels (bit-identical)
    w0, w1 = torch.zeros((2, Hp, Wp, 3), device="cuda"), torch.zeros((2, Hp, Wp, 3), device="cuda")
    _ck(lib.vfi_warp_m2m(ptr(d0d[..., 2:]), 8, 1, ptr(d0d), 8, ptr(w0), 3, 2, Hp, Wp, 3, None), "warp_m2m")
    _ck(lib.vfi_m2m_warp_image4(ptr(img4), ptr(d0d), 8, ptr(w1), 3, Hp, Wp, None), "warp_image4")
    torch.cuda.synchronize()
    assert torch.equal(w0, w1), describe_diff(w1.cpu(), w0.cpu(), "warp_image4 vs warp_m2m")
    tiles = (Hp // 32) * (Wp // 32)
    ranges = torch.full((8, tiles, 4), float("nan"), device="cuda")
    smax = torch.full((8,), float("nan"), device="cuda")
    _ck(lib.vfi_m2m_photo_tiles(ptr(d0d), 8, ptr(rd), 12, 0.7, ptr(img4), ptr(tf1), ptr(e1), ptr(ranges), ptr(smax), Hp, Wp, None), "photo_tiles")
    torch.cuda.synchronize()
    assert torch.equal(tf0, tf1) and torch.equal(e0, e1)
    assert torch.equal(img4[..., :3].cpu(), d0[..., 2:5]) and bool((img4[..., 3] == 1).all())
    want_r, want_s = tile_ranges(tf1.cpu().numpy())
    assert np.array_equal(ranges.cpu().numpy(), want_r) and np.array_equal(smax.cpu().numpy(), want_s)


@pytest.mark.parametrize("h,w", [(100, 150), (270, 480)])
def test_engine_fused_equals_three_step(lib, h, w):
    """The C-side object with the A/B option off and on: the same prepare, the two render forms, bit-identical frames."""
    from cfi_amd.m2m import M2MEngine

    fr = synth.smooth_frames(2, h, w, seed=3, shift=4.0)
    eng = M2MEngine(synth.m2m_synth_state_dict(1234))
    try:
        eng.prepare(fr[0].cuda().contiguous(), fr[1].cuda().contiguous())
        outs = {}
        for mode in (1, 0, 1):
            assert lib.vfi_test_set_option(b"m2m_fused", mode) == 0
            outs.setdefault(mode, []).append([eng.render(t).cpu() for t in (0.5, 0.25, 0.8)])
        for a, b in zip(outs[1][0], outs[0][0]):
            assert torch.equal(a, b), describe_diff(a, b, "engine: fused vs three-step")
        for a, b in zip(outs[1][0], outs[1][1]):
            assert torch.equal(a, b), "run-to-run determinism"
    finally:
        lib.vfi_test_set_option(b"m2m_fused", 1)
        eng.close()


@pytest.mark.parametrize("h,w", [(100, 150), (270, 480), (1080, 1920)])
def test_prepare_side_stream_equals_one_stream(lib, h, w):
    """r6 A/B form (option m2m_side, default off: measured neutral): prepare() forks the refinement network's image-pyramid convolutions onto
    the object's side stream beside the PWC flow network.  Same kernels on the same tensors: the frames must be bit-identical to the one-stream order, call after call
    (a second pair re-uses the pyramid's windows while the previous pair's up path has just overwritten them)."""
    from cfi_amd.m2m import M2MEngine

    fr = synth.smooth_frames(3, h, w, seed=3, shift=4.0)
    x = [f.cuda().contiguous() for f in fr]
    eng = M2MEngine(synth.m2m_synth_state_dict(1234))
    try:
        outs = {}
        for mode in (1, 0, 1, 1):
            assert lib.vfi_test_set_option(b"m2m_side", mode) == 0
            got = []
            for a, b in ((0, 1), (1, 2), (2, 0)):
                eng.prepare(x[a], x[b])
                got.append(eng.render(0.5).cpu())
            outs.setdefault(mode, []).append(got)
        for k in range(3):
            assert torch.equal(outs[1][0][k], outs[0][0][k]), describe_diff(outs[1][0][k], outs[0][0][k], f"side stream vs one stream, pair {k}")
            assert torch.equal(outs[1][0][k], outs[1][1][k]) and torch.equal(outs[1][0][k], outs[1][2][k]), "run-to-run determinism"
    finally:
        lib.vfi_test_set_option(b"m2m_side", 0)
        eng.close()


# ---- the generic 4-channel splat on the same machinery (softsplat4_kernel: classification + compaction) ------------------------------
@pytest.mark.parametrize("N,H,W,kind", [(1, 96, 160, "smooth"), (2, 70, 90, "translate"), (1, 136, 240, "noise8"), (1, 64, 96, "noise1"), (1, 128, 128, "zoom"),
                                        (1, 64, 64, "point"), (1, 200, 300, "far")])
def test_softsplat4_vs_c_oracle_and_list_kernel(lib, N, H, W, kind):
    """vfi_softsplat_sum, C = 4: the staged list gather with source compaction (option splat_atomic = 3: the M2M render kernel's
    machinery as one splat) against the plain-C oracle, and against the default list kernel: bit-identical where neither takes a side
    path (coherent fields), within summation-order noise elsewhere."""
    g = torch.Generator().manual_seed(H * 7 + W)
    a = torch.rand((N, H, W, 4), generator=g)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    f = torch.empty((N, H, W, 2))
    for n in range(N):
        if kind == "smooth":
            f[n, ..., 0], f[n, ..., 1] = 3.0 * torch.sin(yy / 37.0 + n) + 1.3, 2.0 * torch.cos(xx / 41.0) - 0.6
        elif kind == "translate":
            f[n, ..., 0], f[n, ..., 1] = 2.25 + n, -1.75
        elif kind == "noise8":
            f[n] = torch.randn((H, W, 2), generator=g) * 8.0
        elif kind == "noise1":
            f[n] = torch.randn((H, W, 2), generator=g)
        elif kind == "zoom":
            f[n, ..., 0], f[n, ..., 1] = (W / 2 - xx) * 0.7, (H / 2 - yy) * 0.7
        elif kind == "point":
            f[n, ..., 0], f[n, ..., 1] = (W / 3 + 0.37) - xx, (H / 2 + 0.21) - yy
        elif kind == "far":
            f[n, ..., 0], f[n, ..., 1] = 140.0 * torch.sin(yy / 60.0), -90.0 * torch.cos(xx / 80.0)
    if kind == "noise8":
        f[0, 3, 4, 0] = float("nan")
        f[0, 2, 2, 1] = float("-inf")
    ad, fd = a.cuda(), f.cuda()

    def run(mode):
        assert lib.vfi_test_set_option(b"splat_atomic", mode) == 0
        try:
            out = torch.full((N, H, W, 4), float("nan"), device="cuda")
            _ck(lib.vfi_softsplat_sum(ptr(ad), ptr(fd), ptr(out), N, H, W, 4, None), "softsplat")
            torch.cuda.synchronize()
            return out.cpu()
        finally:
            lib.vfi_test_set_option(b"splat_atomic", 0)

    got, old = run(3), run(0)
    want = torch.from_numpy(M.softsplat_sum(np.ascontiguousarray(a.permute(0, 3, 1, 2).numpy()), np.ascontiguousarray(f.permute(0, 3, 1, 2).numpy()))).permute(0, 2, 3, 1)
    tol = 2e-5 * max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= tol, describe_diff(got, want, f"softsplat4 vs oracle {kind}")
    assert torch.equal(got, run(3)), "run-to-run determinism"
    if kind in ("smooth", "translate", "noise1"):
        assert torch.equal(got, old), describe_diff(got, old, f"softsplat4 vs list kernel {kind}")
    if kind == "translate":
        assert torch.equal(got, want), describe_diff(got, want, "uniform translation: the sequential oracle's order")
