"""M2M VFI node — host-side mirror of the reference's ``M2M_VFI`` over the HIP library.

Node shape and frame loop follow vfi_models/m2m/__init__.py:14-60 + vfi_utils.generic_frame_loop
(vfi_utils.py:149-389, timestep mode): frame_i, its multiplier-1 middle frames, ..., last frame; skipped pairs keep
their first frame; no clamp.  The model (vfi_models/m2m/M2M_arch.py ``M2M_PWC.forward`` :894-1037) is executed as
a sequence of C-ABI calls:

    Basic("...sconv(2)-prelu-conv(3,replpad)-prelu...") / conv() / Conv2 / deconv()   -> vfi_conv_forward_ex (MFMA)
    costvol_func / softsplat_func (the reference's cupy kernels)                        -> vfi_costvol9x9 / vfi_softsplat_sum
    backwarp (grid_sample zeros, align_corners=True)                                    -> vfi_warp_m2m
    F.interpolate(bilinear) * scale                                                     -> vfi_resize_bilinear
    padding + joint normalisation, cube attention, photometric metric, splat
    inputs, normalise / hole fill / de-normalise                                        -> vfi_m2m_*

Both directions of a pair share every weight, so they run as a batch of 2 (image 0 = frame0->frame1 quantities,
image 1 = the reverse); "partner" reads use the swap flag of the kernels.  torch.cat along channels never
materialises: producers write into channel windows of pre-allocated NHWC tensors.

Unlike the reference (which re-runs the whole network for every timestep, vfi_utils.py:201-211), the flow / refine
part is timestep independent (M2M_arch.py:936-958) and runs ONCE per pair (``prepare``); only the splat runs per
timestep (``render``).  Results are identical; multiplier m costs 1 network pass + (m-1) splats.
"""
import ctypes as C
import typing

import torch

from . import _lib
from .ckpt import cached_engine, load_file_from_github_release
from .dist import all_gather_frames, world
from .m2m_spec import check_state_dict
from .schedule import InterpolationStateList, generic_output_plan, shard_tasks

MODEL_TYPE = "m2m"
RATIO = 4        # M2M_PWC.forward default ratio (the node never overrides it, vfi_models/m2m/__init__.py:51-55)
DEC_CS = 120     # decoder input window: [feature 32 | cost volume 81 | flow 2 | pad] (115 -> x8)
FLOW_OFF = 113


def _p(t, off=0):
    return t.data_ptr() + 4 * off


class _Layer:
    """One conv / transposed conv of the checkpoint, packed for the MFMA kernels."""

    def __init__(self, lib, w, b, kind=0, stride=1, pad_mode=0, chan_map=None, cin_phys=None, prelu=None):
        w = w.detach().to("cpu", torch.float32).contiguous()
        b = b.detach().to("cpu", torch.float32).contiguous()
        if kind == 0:
            cout, cin, k, _ = w.shape
        else:
            cin, cout, k, _ = w.shape
        self.cout, self.cin_phys = cout, cin_phys or (cin + 7) // 8 * 8
        cm = (C.c_int * cin)(*chan_map) if chan_map is not None else None
        pr = prelu.detach().to("cpu", torch.float32).contiguous() if prelu is not None else None
        assert pr is None or pr.numel() == cout
        self.lib, self.kind, self.stride = lib, kind, stride
        self.h = lib.vfi_conv_create_ex(kind, w.data_ptr(), b.data_ptr(), cout, cin, k, stride, pad_mode, cm, self.cin_phys,
                                        pr.data_ptr() if pr is not None else None)
        if not self.h:
            raise RuntimeError("vfi_conv_create_ex failed: " + _lib.last_error())

    def __call__(self, src, soff, dst, doff, act=0, slope=0.0, res=None, roff=0):
        """src/dst: contiguous [N,H,W,C] device tensors; *off = first channel of the window."""
        n, hin, win, cs = src.shape
        s = self.stride
        want = (hin * 2, win * 2) if self.kind == 1 else (hin // s, win // s)
        assert tuple(dst.shape[1:3]) == want and dst.shape[0] == n, (src.shape, dst.shape, want)
        _lib.check(self.lib.vfi_conv_forward_ex(self.h, _p(src, soff), cs, hin, win, _p(dst, doff), dst.shape[-1], n, act, slope,
                                                0.0, 0.0, _p(res, roff) if res is not None else None,
                                                res.shape[-1] if res is not None else 0, _lib.stream_ptr()), "vfi_conv_forward_ex")

    def close(self):
        if self.h:
            self.lib.vfi_conv_destroy(self.h)
            self.h = None


class M2MEngine:
    """Device-resident M2M interpolator: ``prepare(frame0, frame1)`` once per pair, ``render(t)`` per timestep."""

    def __init__(self, state_dict, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("M2M VFI (HIP): no GPU visible; this node has no CPU fallback")
        self.lib = lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _lib.check(lib.vfi_init(self.device.index or 0), "vfi_init")
        check_state_dict(state_dict)
        sd = state_dict
        self.layers = []

        def L(*a, **k):
            l = _Layer(lib, *a, **k)
            self.layers.append(l)
            return l

        def slope(key):
            return float(sd[key].reshape(-1)[0])

        # PWC extractor: 3 x (sconv(2)-prelu, conv(3,replpad)-prelu, conv(3,replpad)-prelu), M2M_arch.py:415-446
        self.ext = []
        for name in ("netOne", "netTwo", "netThr"):
            p = f"netFlow.netExtractor.{name}.netMain."
            self.ext.append([
                (L(sd[p + "0.weight"], sd[p + "0.bias"], stride=2), slope(p + "1.weight")),
                (L(sd[p + "2.weight"], sd[p + "2.bias"], pad_mode=1), slope(p + "3.weight")),
                (L(sd[p + "4.weight"], sd[p + "4.bias"], pad_mode=1), slope(p + "5.weight")),
            ])
        # PWC decoders, level index 0..4 = netOne..netFiv (:449-503)
        self.dec = []
        for name in ("netOne", "netTwo", "netThr", "netFou", "netFiv"):
            p = f"netFlow.{name}.netMain.netMain."
            convs = []
            for i in range(6):
                convs.append((L(sd[p + f"{2 * i}.weight"], sd[p + f"{2 * i}.bias"], pad_mode=1, cin_phys=DEC_CS if i == 0 else None),
                              slope(p + f"{2 * i + 1}.weight") if i < 5 else 0.0))
            self.dec.append(convs)

        def cp(p, stride=1, **k):  # conv() helper: Conv2d(3, stride, 1) + PReLU(cout), :589-602
            return L(sd[p + ".0.weight"], sd[p + ".0.bias"], stride=stride, prelu=sd[p + ".1.weight"], **k)

        q = "MRN.img_pyramid."
        self.pyr = [(cp(q + f"conv{i}.conv1", 2, **({"chan_map": [2, 3, 4], "cin_phys": 8} if i == 1 else {})), cp(q + f"conv{i}.conv2"))
                    for i in range(1, 5)]
        q = "MRN.motion_encdec."
        self.down = [(cp(q + f"down{i}.conv1", 2), cp(q + f"down{i}.conv2")) for i in range(4)]
        self.up = [L(sd[q + f"up{i}.0.weight"], sd[q + f"up{i}.0.bias"], kind=1, stride=2, prelu=sd[q + f"up{i}.1.weight"]) for i in range(4)]
        # conv (8 flow residuals) and conv_m (mask logit) read the same tensor: one layer with 9 outputs (:838-846)
        self.head = L(torch.cat([sd[q + "conv.weight"], sd[q + "conv_m.weight"]], 0), torch.cat([sd[q + "conv.bias"], sd[q + "conv_m.bias"]], 0))
        self.cube = [L(sd[q + f"conv_{n}.1.weight"], sd[q + f"conv_{n}.1.bias"]) for n in ("C", "H", "W")]
        self.alpha = float(sd["paramAlpha"].reshape(-1)[0])
        self.shape = None
        self.prepared = False

    def close(self):
        for l in self.layers:
            l.close()
        self.layers = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------
    def _z(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.device)

    def _alloc(self, H, W):
        if self.shape == (H, W):
            return
        m = RATIO * 16
        Hp, Wp = (H + m - 1) // m * m, (W + m - 1) // m * m
        self.Hp, self.Wp = Hp, Wp
        h, w = Hp // 2, Wp // 2
        z = self._z
        self.d0 = z(2, Hp, Wp, 8)                 # [flow 2 | normalised image 3 | warped partner image 3]
        self.stats = z(2)
        self.ws = torch.zeros(16384, dtype=torch.uint8, device=self.device)
        self.imh = z(2, h, w, 8)                  # half-resolution images for the flow network
        self.dech = [(h >> (l + 1), w >> (l + 1)) for l in range(5)]
        self.decb = [z(2, hh, ww, DEC_CS) for hh, ww in self.dech]
        self.flow = [z(2, hh, ww, 2) for hh, ww in self.dech]
        e = [(Hp >> (l + 1), Wp >> (l + 1)) for l in range(4)]
        self.ench = e
        self.enc = [z(2, e[l][0], e[l][1], 96 << l) for l in range(4)]   # [s | c | warp(partner s,c)], later [s | x]
        self.fl = [None] + [z(2, e[l][0], e[l][1], 2) for l in range(4)]  # flows at 1/2 .. 1/16
        self.s3 = z(2, e[3][0], e[3][1], 256)
        self.pc, self.ph, self.pw = z(2, 1, 1, 256), z(2, e[3][0], 1, 256), z(2, 1, e[3][1], 256)
        self.cc, self.ch, self.cw = z(2, 1, 1, 4096), z(2, e[3][0], 1, 16), z(2, 1, e[3][1], 16)
        self.xf = z(2, Hp, Wp, 16)
        self.r = z(2, Hp, Wp, 12)                 # [8 flow residuals | mask logit | pad]
        self.tf = z(8, Hp, Wp, 2)
        self.e = z(8, Hp, Wp)
        self.sin = z(8, Hp, Wp, 4)
        self.sfl = z(8, Hp, Wp, 2)
        self.sout = z(8, Hp, Wp, 4)
        self.scratch = {}
        self.shape = (H, W)

    def release_workspace(self):
        """Drop the activations; the packed weights stay on the device."""
        for name in ("d0", "imh", "decb", "flow", "enc", "fl", "s3", "xf", "r", "tf", "e", "sin", "sfl", "sout", "_keep"):
            setattr(self, name, None)
        self.scratch = {}
        self.shape = None
        self.prepared = False

    def _tmp(self, name, h, w, c):
        key = (name, h, w, c)
        if key not in self.scratch:
            self.scratch[key] = self._z(2, h, w, c)
        return self.scratch[key]

    def _resize(self, src, soff, dst, doff, c, mul):
        _lib.check(self.lib.vfi_resize_bilinear(_p(src, soff), src.shape[-1], _p(dst, doff), dst.shape[-1], src.shape[0], src.shape[1],
                                                src.shape[2], dst.shape[1], dst.shape[2], c, mul, _lib.stream_ptr()), "vfi_resize_bilinear")

    def _warp(self, src, soff, c, flow, foff, dst, doff, swap=1):
        _lib.check(self.lib.vfi_warp_m2m(_p(src, soff), src.shape[-1], swap, _p(flow, foff), flow.shape[-1], _p(dst, doff), dst.shape[-1],
                                         src.shape[0], src.shape[1], src.shape[2], c, _lib.stream_ptr()), "vfi_warp_m2m")

    # ------------------------------------------------------------------------------------------------
    def prepare(self, frame0, frame1):
        """frame0/frame1: [H,W,C>=3] fp32 device tensors.  Runs everything that does not depend on the timestep."""
        lib, st = self.lib, _lib.stream_ptr()
        H, W, Cc = frame0.shape
        assert frame1.shape == frame0.shape and Cc >= 3 and frame0.is_contiguous() and frame1.is_contiguous()
        self._alloc(H, W)
        Hp, Wp = self.Hp, self.Wp
        self.hw = (H, W)
        self._keep = (frame0, frame1)
        _lib.check(lib.vfi_m2m_normalize(frame0.data_ptr(), frame1.data_ptr(), Cc, H, W, Hp, Wp, self.d0.data_ptr(), 8, 2,
                                         self.stats.data_ptr(), self.ws.data_ptr(), self.ws.numel(), st), "vfi_m2m_normalize")
        # ---- flow network on half-resolution images (:936-939, bidir :521-546)
        self._resize(self.d0, 2, self.imh, 0, 3, 1.0)
        src, soff = self.imh, 0
        for k, stage in enumerate(self.ext):
            hh, ww = self.dech[k]
            a, b = self._tmp("ea", hh, ww, 32), self._tmp("eb", hh, ww, 32)
            stage[0][0](src, soff, a, 0, 1, stage[0][1])
            stage[1][0](a, 0, b, 0, 1, stage[1][1])
            stage[2][0](b, 0, self.decb[k], 0, 1, stage[2][1])
            src, soff = self.decb[k], 0
        for k in (3, 4):  # netFou / netFiv features: avg_pool2d(2, 2) (:438-444)
            _lib.check(lib.vfi_avgpool2(_p(self.decb[k - 1]), DEC_CS, _p(self.decb[k]), DEC_CS, 2, *self.dech[k - 1], 32, st), "vfi_avgpool2")
        for k in (4, 3, 2, 1, 0):
            hh, ww = self.dech[k]
            d = self.decb[k]
            if k == 4:
                _lib.check(lib.vfi_costvol9x9(_p(d), DEC_CS, _p(d), DEC_CS, 1, _p(d), 2, hh, ww, 32, DEC_CS, 32, st), "vfi_costvol9x9")
            else:
                self._resize(self.flow[k + 1], 0, d, FLOW_OFF, 2, 2.0)
                wb = self._tmp("wb", hh, ww, 32)
                self._warp(d, 0, 32, d, FLOW_OFF, wb, 0)
                _lib.check(lib.vfi_costvol9x9(_p(d), DEC_CS, _p(wb), 32, 0, _p(d), 2, hh, ww, 32, DEC_CS, 32, st), "vfi_costvol9x9")
            a, b = self._tmp("da", hh, ww, 128), self._tmp("db", hh, ww, 128)
            cv = self.dec[k]
            cv[0][0](d, 0, a, 0, 1, cv[0][1])
            cv[1][0](a, 0, b, 0, 1, cv[1][1])
            cv[2][0](b, 0, a, 0, 1, cv[2][1])
            cv[3][0](a, 0, b, 0, 1, cv[3][1])
            cv[4][0](b, 0, a, 0, 1, cv[4][1])
            if k == 4:
                cv[5][0](a, 0, self.flow[k], 0, 0)
            else:
                cv[5][0](a, 0, self.flow[k], 0, 0, 0.0, d, FLOW_OFF)
        # ---- motion refinement (MotionRefineNet.forward :866-890, EncDec.forward :718-848)
        self._resize(self.flow[0], 0, self.d0, 0, 2, float(RATIO))
        enc, e = self.enc, self.ench
        coff = [32, 64, 128, 256]          # channel offset of the image-pyramid feature c[l] inside enc[l]
        src, soff = self.d0, 0
        for l in range(4):
            t = self._tmp("pa", e[l][0], e[l][1], 16 << l)
            self.pyr[l][0](src, soff, t, 0, 3)
            self.pyr[l][1](t, 0, enc[l], coff[l], 3)
            src, soff = enc[l], coff[l]
        self._warp(self.d0, 2, 3, self.d0, 0, self.d0, 5)
        src, flow_src = self.d0, self.d0
        for l in range(4):
            t = self._tmp("dn", e[l][0], e[l][1], 32 << l)
            self.down[l][0](src, 0, t, 0, 3)
            dst = enc[l] if l < 3 else self.s3
            self.down[l][1](t, 0, dst, 0, 3)
            self._resize(flow_src, 0, self.fl[l + 1], 0, 2, 0.5)
            flow_src = self.fl[l + 1]
            if l == 3:
                self._cube()
            nfeat = 48 << l                # s + c channels of this level
            self._warp(enc[l], 0, nfeat, self.fl[l + 1], 0, enc[l], nfeat)
            src = enc[l]
        # up path: x is written over the (already consumed) c / warp slots of the level above -> cat(s, x) is a window
        self.up[0](enc[3], 0, enc[2], 128, 3)
        self.up[1](enc[2], 0, enc[1], 64, 3)
        self.up[2](enc[1], 0, enc[0], 32, 3)
        self.up[3](enc[0], 0, self.xf, 0, 3)
        self.head(self.xf, 0, self.r, 0, 0)
        _lib.check(lib.vfi_m2m_photo(_p(self.d0), 8, _p(self.r), 12, self.alpha, _p(self.tf), _p(self.e), Hp, Wp, st), "vfi_m2m_photo")
        self.prepared = True

    def _cube(self):
        lib, st = self.lib, _lib.stream_ptr()
        h, w = self.ench[3]
        for mode, dst in ((1, self.ph), (2, self.pw)):
            _lib.check(lib.vfi_pool_mean(_p(self.s3), 256, _p(dst), 256, 2, h, w, 256, mode, st), "vfi_pool_mean")
        # global mean = mean over rows of the row means (every row has w pixels): 68 values per channel instead of 8160
        _lib.check(lib.vfi_pool_mean(_p(self.ph), 256, _p(self.pc), 256, 2, h, 1, 256, 0, st), "vfi_pool_mean")
        self.cube[0](self.pc, 0, self.cc, 0, 4)
        self.cube[1](self.ph, 0, self.ch, 0, 4)
        self.cube[2](self.pw, 0, self.cw, 0, 4)
        _lib.check(lib.vfi_m2m_cube_apply(_p(self.s3), 256, _p(self.cc), _p(self.ch), 16, _p(self.cw), 16, _p(self.enc[3]), 768, 2, h, w,
                                          256, st), "vfi_m2m_cube_apply")

    def render(self, t, out=None):
        """One middle frame at time t for the prepared pair -> [H,W,3] device tensor (not clamped, like the reference)."""
        assert self.prepared
        lib, st = self.lib, _lib.stream_ptr()
        H, W = self.hw
        Hp, Wp = self.Hp, self.Wp
        if out is None:
            out = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
        _lib.check(lib.vfi_m2m_splat_inputs(_p(self.d0), 8, _p(self.tf), _p(self.e), float(t), _p(self.sin), _p(self.sfl), Hp, Wp, st),
                   "vfi_m2m_splat_inputs")
        _lib.check(lib.vfi_softsplat_sum(_p(self.sin), _p(self.sfl), _p(self.sout), 8, Hp, Wp, 4, st), "vfi_softsplat_sum")
        _lib.check(lib.vfi_m2m_combine(_p(self.sout), _p(self.d0), 8, _p(self.stats), float(t), out.data_ptr(), Hp, Wp, H, W, st),
                   "vfi_m2m_combine")
        return out

    def forward(self, frame0, frame1, t):
        self.prepare(frame0, frame1)
        return self.render(t)


def _load_state_dict(path):
    sd = torch.load(path, map_location="cpu", weights_only=False)
    return sd.state_dict() if hasattr(sd, "state_dict") else sd


def run_plan(engine, frames, plan, tasks, name="M2M VFI"):
    """Shared by the node and the tests.  frames: [N,H,W,C] host tensor; plan/tasks from generic_output_plan.

    Pairs are independent: the (pair, timesteps) tasks are block-partitioned over ranks and the new frames
    all-gathered.  Host side (hostpipe.py): every needed frame is uploaded once through pinned staging ahead of the
    compute stream; new frames and pass-through frames land in their final rows of the output tensor in the background."""
    if not plan:  # list multiplier of zeros: the reference fails in torch.cat of an empty list (vfi_utils.py:386)
        raise RuntimeError(f"{name}: every frame pair was dropped (multiplier 0 everywhere) - nothing to output")
    dev = engine.device
    frames = frames[..., :3]
    H, W = frames.shape[1:3]
    rank, ws = world()
    lo, hi = shard_tasks(tasks, rank, ws)
    counts = [sum(len(ts) for _, ts in tasks[slice(*shard_tasks(tasks, r, ws))]) for r in range(ws)]
    if dev.type != "cuda":  # stand-in engines of the CPU tests: same control flow without the device pipeline
        local = torch.empty((counts[rank], H, W, 3), dtype=torch.float32, device=dev)
        pos = 0
        for pair, ts in tasks[lo:hi]:
            engine.prepare(frames[pair].to(dev, torch.float32).contiguous(), frames[pair + 1].to(dev, torch.float32).contiguous())
            for t in ts:
                engine.render(t, local[pos])
                pos += 1
        new = all_gather_frames(local, counts).cpu()
        src = frames.to("cpu", torch.float32)
        out = torch.empty((len(plan), H, W, 3), dtype=torch.float32)
        for i, (kind, idx) in enumerate(plan):
            out[i] = src[idx] if kind == "src" else new[idx]
        return out

    from .hostpipe import OutputWriter, Uploader
    main = torch.cuda.current_stream(dev)
    wr = OutputWriter(len(plan), H, W, dev)
    new_row = {}
    for i, (kind, idx) in enumerate(plan):
        if kind == "src":
            wr.put_host(i, frames[idx])
        else:
            new_row[idx] = i
    mine = tasks[lo:hi]
    order = sorted({f for pair, _ in mine for f in (pair, pair + 1)})
    up = Uploader(frames, order, dev, main, depth=min(4, len(order)) or 1)
    item_of = {f: i for i, f in enumerate(order)}
    local = torch.empty((counts[rank], H, W, 3), dtype=torch.float32, device=dev)
    first_new = sum(counts[:rank])
    try:
        pos, held, released = 0, {}, 0
        for pair, ts in mine:
            for f in (pair, pair + 1):
                if f not in held:
                    held[f] = up.get(item_of[f])
            engine.prepare(held[pair], held[pair + 1])
            for t in ts:
                engine.render(t, local[pos])
                if ws == 1:
                    wr.put_dev(new_row[first_new + pos], local[pos])
                pos += 1
            # Release only after the LAST render of the pair: some engines' prepare() keeps references to the ring-slot
            # tensors and render() re-reads them (IFRNet, IFUNet), so the `consumed` event must follow those reads.
            while released < item_of[pair + 1]:       # frames before pair+1 are never needed again (tasks ascend)
                up.release(released)
                held.pop(order[released], None)
                released += 1
        if ws > 1:
            new = all_gather_frames(local, counts)
            for k in range(new.shape[0]):
                wr.put_dev(new_row[k], new[k])
    finally:
        up.close()
    return wr.finish()


class M2M_VFI:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "ckpt_name": (["M2M.pth"],),
                "frames": ("IMAGE",),
                "clear_cache_after_n_frames": ("INT", {"default": 10, "min": 1, "max": 1000}),
                "multiplier": ("INT", {"default": 2, "min": 2, "max": 1000}),
            },
            "optional": {"optional_interpolation_states": ("INTERPOLATION_STATES",)},
        }

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "vfi"
    CATEGORY = "ComfyUI-Frame-Interpolation/VFI"

    def vfi(self, ckpt_name: typing.AnyStr, frames: torch.Tensor, clear_cache_after_n_frames: typing.SupportsInt = 1,
            multiplier: typing.SupportsInt = 2, optional_interpolation_states: InterpolationStateList = None, **kwargs):
        assert len(frames) >= 2, f"VFI model M2M requires at least 2 frames to work with, only found {frames.shape[0]}."
        model_path = load_file_from_github_release(MODEL_TYPE, ckpt_name)
        # (the reference rebuilds M2M_PWC on every call, m2m/__init__.py:43-46; see ckpt.cached_engine)
        engine, cached = cached_engine(MODEL_TYPE, model_path, lambda: M2MEngine(_load_state_dict(model_path)))
        try:
            plan, tasks = generic_output_plan(len(frames), multiplier, optional_interpolation_states)
            return (run_plan(engine, frames, plan, tasks),)
        finally:
            if cached:
                torch.cuda.synchronize(engine.device)
                engine.release_workspace()
            else:
                engine.close()
