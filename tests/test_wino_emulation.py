"""CPU check of the Winograd F(2x2,3x3) kernel's HOST side and ADDRESSING (csrc/conv_wino.hip), no GPU needed.

The weight pack is the library's own (vfi_test_pack_wino3x3 -> pack_wino3x3); the device side is emulated lane by lane in
numpy with the kernel's formulas restated one to one — the DMA slot -> (pixel, channel quad) map with its bank swizzle, the
patch read offsets, the B-fragment reads, the MFMA 32x32x2 operand / accumulator lane layout, the register output transform
and the store addressing — and the result is compared with torch's conv2d.  What this pins: the pack layout against the
kernel's read pattern, the swizzle being a bijection the reads invert, the region / quad / XCD work order covering every
(region, channel block) exactly once, and the epilogue's tile -> pixel map.  What it cannot pin (scheduling, LDS-DMA
semantics) is covered by tests/test_gpu_ops.py::test_conv3x3_winograd_*.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F


def pack(lib, w, cin_p, chan_map=None):
    cout, cin = w.shape[:2]
    cout_p = (cout + 31) // 32 * 32
    out = np.zeros(16 * cin_p * cout_p, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    cm = (C.c_int * cin)(*chan_map) if chan_map is not None else None
    n = lib.vfi_test_pack_wino3x3(w.ctypes.data, cout, cin, cm, cin_p, out.ctypes.data, out.size)
    assert n == out.size
    return out, cout_p


def key(py):
    return (py >> 1) & 3


class Geom:
    def __init__(self, rtx):
        self.RTX, self.RTY = rtx, 32 // rtx
        self.RW, self.RH = 2 * rtx, 2 * (32 // rtx)
        self.PW, self.PH = self.RW + 2, self.RH + 2
        self.NITEM = self.PW * self.PH * 2
        self.NA = (self.NITEM + 63) // 64


def work_items(g, N, H, W, NY, grid):
    """(quad, nb) lists per workgroup in the kernel's XCD-aware order (conv_wino_kernel::item)."""
    rx, ry = -(-W // g.RW), -(-H // g.RH)
    R = N * rx * ry
    NQ = -(-R // 4)
    out = []
    for wg in range(grid):
        x, slot, S = wg & 7, wg >> 3, grid >> 3
        nqx = (NQ - x + 7) >> 3 if x < NQ else 0
        items, i = [], 0
        while True:
            jl = slot + i * S
            if jl >= nqx * NY:
                break
            items.append((x + 8 * (jl // NY), jl % NY))
            i += 1
        out.append(items)
    return out, rx, ry, R, NQ


def emulate(x, wp, bias, cout, cout_p, rtx, grid=16, replicate=False):
    """x [N,H,W,in_cs] float32 (Cin_p = in_cs), wp = packed weights; returns [N,H,W,cout] float64-accumulated emulation."""
    g = Geom(rtx)
    N, H, W, cin_p = x.shape
    C8, NY = cin_p // 8, cout_p // 32
    out = np.full((N, H, W, cout), np.nan, np.float64)
    written = np.zeros((N, H, W, cout), np.int32)
    items, rx, ry, R, NQ = work_items(g, N, H, W, NY, grid)
    seen = set()
    lanes = np.arange(64)
    half, l31 = lanes >> 5, lanes & 31
    ty, tx = l31 // g.RTX, l31 % g.RTX
    # patch read bases (floats): pb[(dx & 1) * 2 + (dy >> 1)]
    base = (2 * ty * g.PW + 2 * tx) * 2
    pb = {(dxp, kk): (base + ((half + 2 * dxp) ^ ((ty + kk) & 3))) * 4 for dxp in range(2) for kk in range(2)}
    for wg_items in items:
        for (quad, nb) in wg_items:
            for wave in range(4):
                rg = quad * 4 + wave
                assert (rg, nb) not in seen
                seen.add((rg, nb))
                if rg >= R:
                    continue
                n, rem = divmod(rg, rx * ry)
                ryi, rxi = divmod(rem, rx)
                Ry0, Rx0 = ryi * g.RH, rxi * g.RW
                acc = np.zeros((16, 32, 32), np.float64)      # [xi][tile m][channel n]
                for k in range(C8):
                    # ---- LDS A image of this wave: NA pieces of 64 slots x 4 floats, filled by the DMA
                    A = np.zeros((g.NA * 64, 4), np.float32)
                    for i in range(g.NA):
                        for lane in range(64):
                            sp = i * 64 + lane
                            if sp >= g.NITEM:
                                continue
                            py = (sp >> 1) // g.PW
                            s = sp ^ key(py)
                            pix, q = s >> 1, s & 1
                            px = pix - py * g.PW
                            iy, ix = Ry0 - 1 + py, Rx0 - 1 + px
                            cy, cx = min(max(iy, 0), H - 1), min(max(ix, 0), W - 1)
                            if replicate or (cy == iy and cx == ix):
                                A[sp] = x[n, cy, cx, k * 8 + q * 4:k * 8 + q * 4 + 4]
                    # ---- B image: 16 pieces of 256 floats, verbatim from the pack
                    off = ((nb * C8 + k) * 16) * 256
                    B = wp[off:off + 16 * 256].reshape(16, 64, 4)            # [piece = j*4+xq][lane][xr]
                    # ---- patch reads
                    Af = A.reshape(-1)
                    P = np.zeros((16, 64, 4), np.float32)
                    for dy in range(4):
                        for dx in range(4):
                            o = pb[(dx & 1, dy >> 1)] + (((dy * g.PW + dx) * 2) & ~3) * 4
                            assert (o % 4 == 0).all() and o.max() + 4 <= Af.size
                            P[dy * 4 + dx] = Af[o[:, None] + np.arange(4)[None]]
                    for j in range(4):
                        d = P[:, :, j].reshape(4, 4, 64).astype(np.float32)      # [dy][dx][lane]
                        t = np.stack([d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]])           # rows
                        V = np.stack([t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]], axis=1)  # [r][c][lane]
                        V = V.reshape(16, 64)
                        for xi in range(16):
                            bq = B[j * 4 + (xi >> 2), :, xi & 3]          # per lane
                            # v_mfma_f32_32x32x2_f32: D[m][n] += sum_k A[m][k] B[k][n]; lane (m | n = l31, k = half)
                            for hf in range(2):
                                a_op = V[xi, hf * 32:(hf + 1) * 32].astype(np.float64)     # indexed by m
                                b_op = bq[hf * 32:(hf + 1) * 32].astype(np.float64)        # indexed by n
                                acc[xi] += a_op[:, None] * b_op[None, :]
                # ---- epilogue: lane (l31 = channel, half), register r -> tile m = 8*(r>>2) + 4*half + (r&3)
                for hf in range(2):
                    for r in range(16):
                        m0 = 8 * (r >> 2) + (r & 3)
                        tyy, txx0 = m0 // g.RTX, m0 % g.RTX
                        m = m0 + 4 * hf
                        M = acc[:, m, :].reshape(4, 4, 32)                  # [row][col][channel]
                        s0 = M[0] + M[1] + M[2]
                        s1 = M[1] - M[2] - M[3]
                        y = [s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3], s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]]
                        for ey in range(2):
                            oy = Ry0 + 2 * tyy + ey
                            if oy >= H:
                                continue
                            for ex in range(2):
                                xi_ = (txx0 >> 3) * 8 + 2 * (txx0 & 7) + ex
                                xx = (xi_ >> 3) * 16 + 8 * hf + (xi_ & 7)
                                assert xx == 2 * (m % g.RTX) + ex and tyy == m // g.RTX
                                if Rx0 + xx >= W:
                                    continue
                                for c32 in range(32):
                                    co = nb * 32 + c32
                                    if co < cout:
                                        out[n, oy, Rx0 + xx, co] = y[ey * 2 + ex][c32] + bias[co]
                                        written[n, oy, Rx0 + xx, co] += 1
    assert len(seen) >= R * NY
    assert (written == 1).all(), "every output written exactly once"
    return out


@pytest.mark.parametrize("rtx,N,H,W,cin,cout,replicate", [(8, 2, 11, 21, 12, 40, False), (16, 1, 7, 37, 8, 32, False), (8, 1, 9, 17, 8, 33, True)])
def test_emulated_kernel_matches_conv2d(hip_lib, rtx, N, H, W, cin, cout, replicate):
    g = torch.Generator().manual_seed(rtx * 100 + H)
    cin_p = (cin + 7) // 8 * 8
    x = torch.rand(N, cin, H, W, generator=g, dtype=torch.float64) * 2 - 1
    w = (torch.rand(cout, cin, 3, 3, generator=g, dtype=torch.float64) * 2 - 1) / (cin * 9) ** 0.5
    b = torch.rand(cout, generator=g, dtype=torch.float64) - 0.5
    # physical channels in reverse order (chan_map), padded to cin_p with garbage that zero weights must cancel
    cmap = [cin_p - 1 - c for c in range(cin)]
    xin = torch.rand(N, H, W, cin_p, generator=g, dtype=torch.float64)
    xin[..., cmap] = x.permute(0, 2, 3, 1)
    wp, cout_p = pack(hip_lib, w.numpy().astype(np.float32), cin_p, cmap)
    got = emulate(xin.numpy().astype(np.float32), wp, b.numpy(), cout, cout_p, rtx, replicate=replicate)
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate") if replicate else F.pad(x, (1, 1, 1, 1))
    want = (F.conv2d(xp.float().double(), w.float().double()) + b.view(1, -1, 1, 1)).permute(0, 2, 3, 1).numpy()
    err = np.abs(got - want).max()
    assert err <= 2e-6, err


def test_swizzle_is_a_bijection_and_reads_are_conflict_free():
    """The DMA slot map covers every item once; a wave's ds_read_b128 of one patch position (8x4 tiles) touches 16 distinct
    16-byte bank groups within each of the hardware's lane groups (MI355X_MICROARCH.md, LDS table)."""
    g = Geom(8)
    seen = set()
    for sp in range(g.NITEM):
        py = (sp >> 1) // g.PW
        s = sp ^ key(py)
        assert 0 <= s < g.NITEM and (s >> 1) // g.PW == py
        seen.add(s)
    assert len(seen) == g.NITEM
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    groups += [[l + 32 for l in grp] for grp in groups]
    for dy in range(4):
        for dx in range(4):
            for grp in groups:
                quads = set()
                for lane in grp:
                    half, l31 = lane >> 5, lane & 31
                    ty, tx = l31 // 8, l31 % 8
                    base = (2 * ty * g.PW + 2 * tx) * 2
                    slot = base + ((half + 2 * (dx & 1)) ^ ((ty + (dy >> 1)) & 3)) + (((dy * g.PW + dx) * 2) & ~3)
                    quads.add(slot % 16)
                assert len(quads) == 16, (dy, dx, grp)


def test_work_order_keeps_channel_siblings_on_one_xcd():
    g = Geom(8)
    items, rx, ry, R, NQ = work_items(g, 32, 272, 480, 2, 256)
    flat = {}
    for wg, lst in enumerate(items):
        for (quad, nb) in lst:
            flat.setdefault(quad, set()).add(wg & 7)
    assert len(flat) == NQ and all(len(v) == 1 for v in flat.values())        # all Cout/32 siblings of a quad on one XCD
    counts = [len(lst) for lst in items]
    assert max(counts) - min(counts) <= 2 and sum(counts) == NQ * 2


# ---------------------------------------------------------------------------------------------------------------------
# RIFE's lastconv — ConvTranspose2d(c, 24, 4, 2, 1) + PixelShuffle(2) — as ONE 3x3 convolution with 96 output channels
# (csrc/conv_wino.hip: pack_deconv_as_conv3x3): the host rewrite against torch, on the CPU.


@pytest.mark.parametrize("cin,lo,h,w", [(16, 24, 9, 13), (8, 52, 6, 7), (64, 24, 5, 5)])
def test_deconv_as_conv3x3_equals_conv_transpose(hip_lib, cin, lo, h, w):
    import torch
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(cin + lo)
    wt = (torch.rand(cin, lo, 4, 4, generator=g) * 2 - 1).contiguous()
    b = (torch.rand(lo, generator=g) - 0.5).contiguous()
    x = torch.rand(2, cin, h, w, generator=g) * 2 - 1
    w3 = torch.full((4 * lo, cin, 3, 3), float("nan"))
    b3 = torch.full((4 * lo,), float("nan"))
    n = hip_lib.vfi_test_pack_deconv3x3(wt.data_ptr(), b.data_ptr(), cin, lo, w3.data_ptr(), b3.data_ptr(), w3.numel())
    assert n == w3.numel() and not torch.isnan(w3).any() and not torch.isnan(b3).any()
    assert (w3 != 0).sum().item() <= 4 * lo * cin * 4               # four of the nine taps per parity group
    want = F.conv_transpose2d(x.double(), wt.double(), b.double(), 2, 1)      # [2, lo, 2h, 2w]
    got = F.conv2d(x.double(), w3.double(), b3.double(), padding=1)           # [2, 4 lo, h, w]
    for gi in range(4):
        py, px = gi >> 1, gi & 1
        d = (got[:, gi * lo:(gi + 1) * lo] - want[:, :, py::2, px::2]).abs().max().item()
        assert d <= 1e-12, (gi, d)
