"""RIFE arch "4.0" (``sudo_rife4_269.662_testV1_scale1.pth``) on the HIP library, driven op by op.

The 4.7+ architectures run as one fused C-ABI object (``vfi_rife_*``); 4.0 is the odd one out — PReLU convs, no frame
encoder, flow AND mask accumulated over the blocks, a data-dependent doubling of the block scales, and an optional
Contextnet + Unet refinement (vfi_models/rife/rife_arch.py:176-261,278-342,465-732) — so it is executed like FILM and M2M:
every torch call of the reference maps to one entry point of include/vfi_hip.h

    conv() = Conv2d + PReLU(c), deconv() = ConvTranspose2d(4,2,1) + PReLU   -> vfi_conv_forward_ex (fp32 MFMA)
    F.interpolate(bilinear) (* scale)                                          -> vfi_resize_bilinear
    warp() (border, align_corners=True)                                        -> vfi_warp_rife
    +, copies into concat slots                                                -> vfi_axpby
    clamp/pad/timestep plane, |flow| max, sigmoid blend (+ residual) + crop    -> vfi_rife40_prep / vfi_absmax / vfi_rife40_output

and every ``torch.cat`` is a channel window of a pre-allocated NHWC tensor.  The node's ``fast_mode`` / ``ensemble``
widgets land on ``IFNet.forward``'s ``training`` / ``fastmode`` parameters (rife/__init__.py:200-206) and, for this
architecture only, change the result: ``training=False`` enables the scale doubling (:598-607), ``fastmode=False`` the
refinement (:725-730).  The scale list is doubled IN PLACE and therefore stays doubled for the rest of the node call.
"""
import ctypes as C

import torch

from . import _lib
from .rife_spec import check_state_dict

BLOCK_C = (192, 128, 96, 64)


def _p(t, off=0):
    return t.data_ptr() + 4 * off


class _L:
    """conv (+PReLU) / deconv (+PReLU) layer of the checkpoint."""

    def __init__(self, lib, sd, wkey, bkey, pkey=None, kind=0, stride=1, chan_map=None, cin_phys=None):
        w = sd[wkey].detach().to("cpu", torch.float32).contiguous()
        b = sd[bkey].detach().to("cpu", torch.float32).contiguous()
        pr = sd[pkey].detach().to("cpu", torch.float32).contiguous() if pkey else None
        cout, cin = (w.shape[0], w.shape[1]) if kind == 0 else (w.shape[1], w.shape[0])
        self.cin_phys = cin_phys or (cin + 7) // 8 * 8
        cm = (C.c_int * cin)(*chan_map) if chan_map is not None else None
        self.lib, self.kind, self.stride, self.act = lib, kind, stride, 3 if pr is not None else 0
        self.h = lib.vfi_conv_create_ex(kind, w.data_ptr(), b.data_ptr(), cout, cin, w.shape[2], stride, 0, cm, self.cin_phys,
                                        pr.data_ptr() if pr is not None else None)
        if not self.h:
            raise RuntimeError("vfi_conv_create_ex failed: " + _lib.last_error())

    def __call__(self, src, soff, dst, doff, act=None):
        n, hin, win, cs = src.shape
        want = (hin * 2, win * 2) if self.kind == 1 else (hin // self.stride, win // self.stride)
        assert tuple(dst.shape[1:3]) == want and dst.shape[0] == n, (src.shape, dst.shape, want)
        _lib.check(self.lib.vfi_conv_forward_ex(self.h, _p(src, soff), cs, hin, win, _p(dst, doff), dst.shape[-1], n,
                                                self.act if act is None else act, 0.0, 0.0, 0.0, None, 0, _lib.stream_ptr()),
                   "vfi_conv_forward_ex")

    def close(self):
        if self.h:
            self.lib.vfi_conv_destroy(self.h)
            self.h = None


class Rife40Engine:
    def __init__(self, state_dict, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("RIFE VFI (HIP): no GPU visible; this node has no CPU fallback")
        self.lib = lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _lib.check(lib.vfi_init(self.device.index or 0), "vfi_init")
        check_state_dict(state_dict, "4.0")
        sd = state_dict
        self.layers = []

        def cp(p, stride=1, **k):
            l = _L(lib, sd, p + ".0.weight", p + ".0.bias", p + ".1.weight", stride=stride, **k)
            self.layers.append(l)
            return l

        def dc(p, **k):
            l = _L(lib, sd, p + ".0.weight", p + ".0.bias", p + ".1.weight", kind=1, stride=2, **k)
            self.layers.append(l)
            return l

        self.blocks = []
        for b in range(4):
            p = f"block{b}."
            last = _L(lib, sd, p + "lastconv.weight", p + "lastconv.bias", None, kind=1, stride=2)
            self.layers.append(last)
            self.blocks.append((cp(p + "conv0.0", 2, cin_phys=8 if b == 0 else 16), cp(p + "conv0.1", 2),
                                [cp(p + f"convblock.{i}") for i in range(8)], last))
        # Contextnet: the first conv reads the image out of the (img0 | img1 | t) tensor, one layer object per image
        self.ctx1 = [cp("contextnet.conv1.conv1", 2, chan_map=[3 * k, 3 * k + 1, 3 * k + 2], cin_phys=8) for k in range(2)]
        self.ctx = [None, (None, cp("contextnet.conv1.conv2"))] + \
                   [(cp(f"contextnet.conv{i}.conv1", 2), cp(f"contextnet.conv{i}.conv2")) for i in (2, 3, 4)]
        self.down = [(cp("unet.down0.conv1", 2, cin_phys=24), cp("unet.down0.conv2"))] + \
                    [(cp(f"unet.down{i}.conv1", 2), cp(f"unet.down{i}.conv2")) for i in (1, 2, 3)]

        def swap(n):   # reference cat(x, s): x first; physical [s | x]
            return list(range(n, 2 * n)) + list(range(n))

        self.up = [dc("unet.up0"), dc("unet.up1", chan_map=swap(128)), dc("unet.up2", chan_map=swap(64)), dc("unet.up3", chan_map=swap(32))]
        self.uconv = _L(lib, sd, "unet.conv.weight", "unet.conv.bias")
        self.layers.append(self.uconv)
        self.cfg = None

    def close(self):
        for l in self.layers:
            l.close()
        self.layers = []

    # ------------------------------------------------------------------------------------------------
    def _z(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.device)

    def configure(self, H, W, B):
        if self.cfg == (H, W, B):
            return
        torch.cuda.synchronize(self.device)
        self.Hp, self.Wp = (H + 63) // 64 * 64, (W + 63) // 64 * 64
        Hp, Wp, z = self.Hp, self.Wp, self._z
        self.img = z(B, Hp, Wp, 8)       # (img0 3 | img1 3 | timestep | 0): block-0 input
        self.xin = z(B, Hp, Wp, 8)       # (warped img0 3 | warped img1 3 | timestep | mask): input of blocks 1..3 (+ flow)
        self.flow = z(B, Hp, Wp, 4)
        self.df = z(B, Hp, Wp, 4)
        self.dm = z(B, Hp, Wp, 1)
        self.amax = z(2)
        self.scratch = {}
        self.cfg = (H, W, B)

    def _tmp(self, name, h, w, c):
        key = (name, h, w, c)
        if key not in self.scratch:
            self.scratch[key] = self._z(self.cfg[2], h, w, c)
        return self.scratch[key]

    def _resize(self, src, soff, dst, doff, c, mul=1.0):
        _lib.check(self.lib.vfi_resize_bilinear(_p(src, soff), src.shape[-1], _p(dst, doff), dst.shape[-1], src.shape[0], src.shape[1],
                                                src.shape[2], dst.shape[1], dst.shape[2], c, mul, _lib.stream_ptr()), "vfi_resize_bilinear")

    def _warp(self, src, soff, c, flow, foff, dst, doff):
        _lib.check(self.lib.vfi_warp_rife(_p(src, soff), src.shape[-1], _p(flow, foff), flow.shape[-1], _p(dst, doff), dst.shape[-1],
                                          src.shape[0], src.shape[1], src.shape[2], c, _lib.stream_ptr()), "vfi_warp_rife")

    def _ax(self, a, aoff, b, boff, out, ooff, c, alpha=1.0, beta=1.0):
        px = a.shape[0] * a.shape[1] * a.shape[2]
        _lib.check(self.lib.vfi_axpby(_p(a, aoff), a.shape[-1], _p(b, boff) if b is not None else None, b.shape[-1] if b is not None else 0,
                                      _p(out, ooff), out.shape[-1], px, c, alpha, beta, _lib.stream_ptr()), "vfi_axpby")

    def _block(self, i, x, n_img, has_flow, scale):
        """IFBlock.forward (rife_arch.py:237-261): results in self.df (flow delta * 2*scale) and self.dm (mask delta)."""
        Hp, Wp = self.Hp, self.Wp
        hs, ws = int(round(Hp / scale)), int(round(Wp / scale))
        if hs * scale != Hp or ws * scale != Wp or hs % 4 or ws % 4:
            raise RuntimeError(f"RIFE 4.0: block scale {scale} does not divide the padded size {Hp}x{Wp} (the reference fails here too)")
        c00, c01, cb, last = self.blocks[i]
        c = BLOCK_C[i]
        xs = self._tmp("xs", hs, ws, c00.cin_phys)
        self._resize(x, 0, xs, 0, n_img)
        if has_flow:
            self._resize(self.flow, 0, xs, 8, 4, 1.0 / scale)
        a = self._tmp("a", hs // 2, ws // 2, c // 2)
        f = self._tmp("f", hs // 4, ws // 4, c)
        p, q = self._tmp("p", hs // 4, ws // 4, c), self._tmp("q", hs // 4, ws // 4, c)
        c00(xs, 0, a, 0)
        c01(a, 0, f, 0)
        cur = f
        for k in range(8):
            nxt = p if k % 2 == 0 else q
            cb[k](cur, 0, nxt, 0)
            cur = nxt
        self._ax(cur, 0, f, 0, p if cur is q else q, 0, c)          # convblock(feat) + feat
        r = p if cur is q else q
        t = self._tmp("t", hs // 2, ws // 2, 8)
        last(r, 0, t, 0)
        self._resize(t, 0, self.df, 0, 4, scale * 2.0)
        self._resize(t, 4, self.dm, 0, 1)

    def _warp_images(self):
        self._warp(self.img, 0, 3, self.flow, 0, self.xin, 0)
        self._warp(self.img, 3, 3, self.flow, 2, self.xin, 3)

    def forward(self, frames0, frames1, timesteps, scale_list, training, fastmode, out):
        """frames0/1: lists of [H,W,C>=3] device tensors, one pair per task; scale_list: the node's LIST (doubled in place when
        the reference would); out [B,H,W,3] device tensor, clamped to [0,1] like the node's output."""
        lib, st = self.lib, _lib.stream_ptr()
        B = len(timesteps)
        H, W = frames0[0].shape[:2]
        assert self.cfg is not None and self.cfg[:2] == (H, W) and self.cfg[2] == B, (self.cfg, H, W, B)
        Hp, Wp = self.Hp, self.Wp
        for b in range(B):
            f0, f1 = frames0[b], frames1[b]
            assert f0.is_cuda and f0.is_contiguous() and f1.is_contiguous() and f0.shape == f1.shape
            _lib.check(lib.vfi_rife40_prep(f0.data_ptr(), f1.data_ptr(), f0.shape[2], H, W, float(timesteps[b]), _p(self.img[b]),
                                           Hp, Wp, st), "vfi_rife40_prep")
        self._ax(self.img, 6, None, 0, self.xin, 6, 1)              # timestep plane of the later blocks' input

        def block0():
            self._block(0, self.img, 7, False, scale_list[0])
            self._ax(self.df, 0, None, 0, self.flow, 0, 4)
            self._ax(self.dm, 0, None, 0, self.xin, 7, 1)
            self._warp_images()

        block0()
        for i in range(1, 4):
            self._block(i, self.xin, 8, True, scale_list[i])
            if i == 1 and not training:
                px = B * Hp * Wp
                _lib.check(lib.vfi_absmax(_p(self.df, 0), 4, 2, px, _p(self.amax, 0), st), "vfi_absmax")
                _lib.check(lib.vfi_absmax(_p(self.df, 2), 4, 2, px, _p(self.amax, 1), st), "vfi_absmax")
                m = self.amax.cpu()
                if float(m[0]) > 32 and float(m[1]) > 32:
                    for k in range(4):
                        scale_list[k] *= 2
                    block0()
                    self._block(1, self.xin, 8, True, scale_list[1])
            self._ax(self.flow, 0, self.df, 0, self.flow, 0, 4)      # flow = flow + f0
            self._ax(self.xin, 7, self.dm, 0, self.xin, 7, 1)        # mask = mask + m0
            self._warp_images()
        res = None
        if not fastmode:
            res = self._refine()
        _lib.check(lib.vfi_rife40_output(_p(self.xin), 8, _p(self.xin, 7), 8, _p(res) if res is not None else None, 4, out.data_ptr(),
                                         B, Hp, Wp, H, W, st), "vfi_rife40_output")
        return out

    def _refine(self):
        """Contextnet on both images + Unet (rife_arch.py:278-342,725-730) -> sigmoid output [B,Hp,Wp,4] (3 used)."""
        Hp, Wp = self.Hp, self.Wp
        hw = [(Hp >> j, Wp >> j) for j in range(5)]
        u = [self._tmp("u0", Hp, Wp, 24)] + [self._tmp(f"u{j}", *hw[j], 64 << (j - 1)) for j in range(1, 5)]
        fl = [self.flow] + [self._tmp(f"fl{j}", *hw[j], 4) for j in range(1, 5)]
        for j in range(1, 5):
            self._resize(fl[j - 1], 0, fl[j], 0, 4, 0.5)
        for k in range(2):
            x = None
            for j in range(1, 5):
                cw = 8 << j                                         # 16, 32, 64, 128 context channels
                t = self._tmp("ct", *hw[j], cw)
                y = self._tmp("cy", *hw[j], cw)
                if j == 1:
                    self.ctx1[k](self.img, 0, t, 0)
                else:
                    self.ctx[j][0](x, 0, t, 0)
                self.ctx[j][1](t, 0, y, 0)
                self._warp(y, 0, cw, fl[j], 2 * k, u[j], 2 * cw + cw * k)   # slots after the 2*cw channels of s_{j-1}
                x = y
        # Unet
        self._ax(self.img, 0, None, 0, u[0], 0, 6)                  # img0, img1
        self._ax(self.xin, 0, None, 0, u[0], 6, 6)                  # warped img0, img1
        self._ax(self.xin, 7, None, 0, u[0], 12, 1)                 # mask (raw, not the sigmoid)
        self._ax(self.flow, 0, None, 0, u[0], 13, 4)
        for j in range(4):
            cw = 32 << j
            t = self._tmp("dt", *hw[j + 1], cw)
            self.down[j][0](u[j], 0, t, 0)
            self.down[j][1](t, 0, u[j + 1], 0)
        self.up[0](u[4], 0, u[3], 128)                              # x over the consumed context slots: [s2 | x]
        self.up[1](u[3], 0, u[2], 64)
        self.up[2](u[2], 0, u[1], 32)
        uf = self._tmp("uf", Hp, Wp, 16)
        self.up[3](u[1], 0, uf, 0)
        res = self._tmp("res", Hp, Wp, 4)
        self.uconv(uf, 0, res, 0, act=4)
        return res


def run_tasks40(engine, frames, tasks, batch_size, scale_list, fast_mode, ensemble, writer, rows):
    """The node's batch loop for arch 4.0 (rife/__init__.py:186-222): ``tasks`` [(pair, t)], new frame i -> writer row rows[i].
    ``fast_mode`` / ``ensemble`` are forwarded the way the reference forwards them (as ``training`` / ``fastmode``)."""
    from .hostpipe import Uploader

    dev = engine.device
    H, W = frames.shape[1:3]
    bs = max(1, int(batch_size))
    order = sorted({f for p, _ in tasks for f in (p, p + 1)})
    item_of = {f: k for k, f in enumerate(order)}
    up = Uploader(frames, order, dev, torch.cuda.current_stream(dev), depth=min(len(order), 2 * bs + 2) or 1)   # a batch over disjoint pairs holds 2*bs frames at once
    held, released = {}, 0
    keep = []
    try:
        for pos in range(0, len(tasks), bs):
            bt = tasks[pos:pos + bs]
            for p, _ in bt:
                for f in (p, p + 1):
                    if f not in held:
                        held[f] = up.get(item_of[f])
            engine.configure(H, W, len(bt))
            out = torch.empty((len(bt), H, W, 3), dtype=torch.float32, device=dev)
            engine.forward([held[p] for p, _ in bt], [held[p + 1] for p, _ in bt], [t for _, t in bt], scale_list, fast_mode, ensemble, out)
            for i in range(len(bt)):
                writer.put_dev(rows[pos + i], out[i])
            keep.append(out)
            last_needed = bt[-1][0]                       # tasks ascend by pair: frames before the last pair are done
            while released < item_of[last_needed]:
                up.release(released)
                held.pop(order[released], None)
                released += 1
    finally:
        up.close()
    return keep
