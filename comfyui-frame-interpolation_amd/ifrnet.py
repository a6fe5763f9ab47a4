"""IFRNet VFI node — host-side mirror of vfi_models/ifrnet/__init__.py over the HIP library.

IRFNet_L / IRFNet_S (IFRNet_L_arch.py, IFRNet_S_arch.py) are plain PReLU-conv encoder/decoder pyramids with border
warps; every torch call of the reference maps to one entry point of include/vfi_hip.h

    convrelu = Conv2d + PReLU(c), ConvTranspose2d(4, 2, 1), ResBlock's residual + PReLU   -> vfi_conv_forward_ex (fp32 MFMA)
    the L model's 7x7 stride-2 head conv (3 input channels)                               -> vfi_conv7x7s2_prelu
    F.pad, mean_ removal                                        -> vfi_ifrnet_prep / vfi_pool_mean / vfi_ifrnet_center
    resize() = F.interpolate(scale_factor=...)                  -> vfi_resize_bilinear_ratio (vfi_resize_bilinear for the x2 steps)
    warp() on features                                          -> vfi_warp_rife
    embt.repeat, torch.sigmoid, +, copies into concat slots     -> vfi_fill_items / vfi_sigmoid / vfi_axpby
    both image warps + mask blend + mean_ + residual + clamp + crop                       -> vfi_ifrnet_output

and every ``torch.cat`` is a channel window of a pre-allocated NHWC tensor (the ConvTranspose2d of decoder k writes
straight into the input tensor of decoder k-1; ``chan_map`` re-orders the weights to that physical layout).  ResBlock's
in-place convs on the last ``side`` channels read a channel window and are copied back.

Reference behaviour kept as is: the node calls ``model(frame_0, frame_1, timestep, scale_factor)`` against
``forward(img0, img1, scale_factor=1.0, timestep=0.5)`` (ifrnet/__init__.py:49-50, IFRNet_L_arch.py:225), so the loop's
timestep k/multiplier is the network's WORKING-RESOLUTION factor and the ``scale_factor`` widget is the time embedding.
Timesteps whose resized frame is not a multiple of 16 fail in the reference's ``torch.cat`` and raise here as well.
"""
import ctypes as C
import math
import typing

import torch

from . import _lib
from .ckpt import begin_call, cached_engine, end_call, load_file_from_github_release
from .lanes import lane_set
from .lanes import configure as configure_lanes
from .ifrnet_spec import CKPT_NAMES, CONFIG, check_state_dict, decoder_io, kind_of
from .schedule import InterpolationStateList, generic_output_plan

MODEL_TYPE = "ifrnet"


def _p(t, off=0):
    return t.data_ptr() + 4 * off


def _cs(c):
    """physical channel stride of a c-channel activation"""
    return (c + 7) // 8 * 8


class _Layer:
    """Conv2d 3x3 (+PReLU) or ConvTranspose2d(4,2,1) of the checkpoint as a vfi_conv layer object."""

    def __init__(self, lib, w, b, slopes=None, kind=0, stride=1, chan_map=None, cin_phys=None):
        w = w.detach().to("cpu", torch.float32).contiguous()
        b = b.detach().to("cpu", torch.float32).contiguous()
        pr = slopes.detach().to("cpu", torch.float32).contiguous() if slopes is not None else None
        cout, cin = (w.shape[0], w.shape[1]) if kind == 0 else (w.shape[1], w.shape[0])
        self.cin_phys = cin_phys or _cs(cin)
        cm = (C.c_int * cin)(*chan_map) if chan_map is not None else None
        self.lib, self.kind, self.stride, self.act, self.cout = lib, kind, stride, 3 if pr is not None else 0, cout
        self.h = lib.vfi_conv_create_ex(kind, w.data_ptr(), b.data_ptr(), cout, cin, w.shape[2], stride, 0, cm, self.cin_phys,
                                        pr.data_ptr() if pr is not None else None)
        if not self.h:
            raise RuntimeError("vfi_conv_create_ex failed: " + _lib.last_error())

    def __call__(self, src, soff, dst, doff, res=None):
        n, hin, win, cs = src.shape
        want = (hin * 2, win * 2) if self.kind == 1 else (hin // self.stride, win // self.stride)
        assert tuple(dst.shape[1:3]) == want and dst.shape[0] == n, (src.shape, dst.shape, want)
        _lib.check(self.lib.vfi_conv_forward_ex(self.h, _p(src, soff), cs, hin, win, _p(dst, doff), dst.shape[-1], n, self.act, 0.0, 0.0,
                                                0.0, _p(res) if res is not None else None, res.shape[-1] if res is not None else 0,
                                                _lib.stream_ptr()), "vfi_conv_forward_ex")

    def close(self):
        if self.h:
            self.lib.vfi_conv_destroy(self.h)
            self.h = None


class IFRNetEngine:
    """IRFNet_L / IRFNet_S forward on the device.  ``forward`` takes N frame pairs that share one working-resolution
    factor; ``prepare`` / ``render`` adapt it to the (pair, timestep) loop shared with the M2M node."""

    def __init__(self, state_dict, kind, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("IFRNet VFI (HIP): no GPU visible; this node has no CPU fallback")
        self.lib = lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _lib.check(lib.vfi_init(self.device.index or 0), "vfi_init")
        check_state_dict(state_dict, kind)
        self.kind, sd = kind, state_dict
        self.widths, self.k0, self.side = CONFIG[kind]
        self.layers = []
        self.embt = 1.0

        def mk(*a, **k):
            l = _Layer(lib, *a, **k)
            self.layers.append(l)
            return l

        def convrelu(p, **k):
            return mk(sd[p + ".0.weight"], sd[p + ".0.bias"], sd[p + ".1.weight"], **k)

        self.enc, self.head = [], None
        for lvl in range(1, 5):
            p = f"encoder.pyramid{lvl}"
            if lvl == 1 and self.k0 == 7:   # IFRNet_L: 7x7 head on 3 channels, its own kernel; weights as [ky][kx][ci][co]
                f32 = lambda t: t.detach().to(self.device, torch.float32).contiguous()   # noqa: E731
                self.head = (f32(sd[p + ".0.0.weight"].permute(2, 3, 1, 0)), f32(sd[p + ".0.0.bias"]), f32(sd[p + ".0.1.weight"]))
                first = None
            else:
                first = convrelu(p + ".0", stride=2, cin_phys=8 if lvl == 1 else None)
            self.enc.append((first, convrelu(p + ".1")))
        self.dec = {}
        for d, din, c, dout in decoder_io(kind):
            p = f"decoder{d}.convblock"
            if d == 4:
                conv0 = convrelu(p + ".0")                     # physical layout = cat(f0_4, f1_4, embt)
            else:
                f = (din - 4) // 3                              # physical [flow0 2 | flow1 2 | ft_ f | f0_warp f | f1_warp f]
                cmap = list(range(4, 4 + 3 * f)) + [0, 1, 2, 3]  # reference order: ft_, f0_warp, f1_warp, up_flow0, up_flow1
                conv0 = convrelu(p + ".0", chan_map=cmap)
            off = c - self.side
            windowed = off % 4 == 0                             # else the side convs read the whole tensor through a chan_map
            sidek = {} if windowed else dict(chan_map=list(range(off, c)), cin_phys=_cs(c))
            self.dec[d] = dict(
                c=c, dout=dout, windowed=windowed, conv0=conv0, c1=convrelu(p + ".1.conv1"), c2=convrelu(p + ".1.conv2", **sidek),
                c3=convrelu(p + ".1.conv3"), c4=convrelu(p + ".1.conv4", **sidek),
                c5=mk(sd[p + ".1.conv5.weight"], sd[p + ".1.conv5.bias"], sd[p + ".1.prelu.weight"]),
                up=mk(sd[p + ".2.weight"], sd[p + ".2.bias"], None, kind=1, stride=2))
        self.scratch = {}
        self._pair = None

    def close(self):
        for l in self.layers:
            l.close()
        self.layers = []
        self.release_workspace()

    def release_workspace(self):
        """Drop the activations; the packed weights stay on the device."""
        self.scratch = {}
        self._pair = None

    def workspace_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.scratch.values())

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 — interpreter shutdown
            pass

    # ------------------------------------------------------------------------------------------------
    def _t(self, name, *shape):
        key = (name,) + tuple(shape)
        if key not in self.scratch:
            self.scratch[key] = torch.zeros(shape, dtype=torch.float32, device=self.device)
        return self.scratch[key]

    def _ax(self, a, aoff, b, boff, out, ooff, c, alpha=1.0, beta=1.0):
        px = a.shape[0] * a.shape[1] * a.shape[2]
        _lib.check(self.lib.vfi_axpby(_p(a, aoff), a.shape[-1], _p(b, boff) if b is not None else None, b.shape[-1] if b is not None else 0,
                                      _p(out, ooff), out.shape[-1], px, c, alpha, beta, _lib.stream_ptr()), "vfi_axpby")

    def _warp(self, src, c, flow, foff, dst, doff):
        _lib.check(self.lib.vfi_warp_rife(_p(src), src.shape[-1], _p(flow, foff), flow.shape[-1], _p(dst, doff), dst.shape[-1],
                                          src.shape[0], src.shape[1], src.shape[2], c, _lib.stream_ptr()), "vfi_warp_rife")

    def _up2(self, src, dst, c):
        """dst[..., :c] = 2.0 * resize(src[..., :c], 2.0)  (the x2 is exact in fp32, so it commutes with the resize)"""
        _lib.check(self.lib.vfi_resize_bilinear(_p(src), src.shape[-1], _p(dst), dst.shape[-1], src.shape[0], src.shape[1], src.shape[2],
                                                dst.shape[1], dst.shape[2], c, 2.0, _lib.stream_ptr()), "vfi_resize_bilinear")

    def _resize(self, src, soff, dst, doff, c, ratio, post):
        _lib.check(self.lib.vfi_resize_bilinear_ratio(_p(src, soff), src.shape[-1], _p(dst, doff), dst.shape[-1], src.shape[0], src.shape[1],
                                                      src.shape[2], dst.shape[1], dst.shape[2], c, ratio, ratio, post, _lib.stream_ptr()),
                   "vfi_resize_bilinear_ratio")

    def _side(self, layer, buf, c, windowed, tmp):
        """buf[..., c-side:c] = convrelu(buf[..., c-side:c])   (ResBlock.forward, IFRNet_L_arch.py:115-121)"""
        layer(buf, c - self.side if windowed else 0, tmp, 0)
        self._ax(tmp, 0, None, 0, buf, c - self.side, self.side)

    def _decoder(self, d, din, target):
        """decoder{d}.convblock: convrelu -> ResBlock -> ConvTranspose2d, the latter written to channels 0.. of ``target``."""
        L = self.dec[d]
        n, h, w, _ = din.shape
        c, cs = L["c"], _cs(L["c"])
        x, a, b, y = (self._t(f"dec_{k}", n, h, w, cs) for k in "xaby")
        tmp = self._t("dec_t", n, h, w, _cs(self.side))
        L["conv0"](din, 0, x, 0)
        L["c1"](x, 0, a, 0)
        self._side(L["c2"], a, c, L["windowed"], tmp)
        L["c3"](a, 0, b, 0)
        self._side(L["c4"], b, c, L["windowed"], tmp)
        L["c5"](b, 0, y, 0, res=x)        # prelu(x + conv5(out))
        L["up"](y, 0, target, 0)

    @staticmethod
    def geometry(H, W, sf):
        """Sizes the reference's F.interpolate calls produce (output size = floor(in * scale_factor), IFRNet_L_arch.py:250-251,
        281-288): padded frame, working resolution, and the size of the final flow / mask / residual fields."""
        Hp, Wp = ((H - 1) // 64 + 1) * 64, ((W - 1) // 64 + 1) * 64
        hs, ws = int(math.floor(float(Hp) * sf)), int(math.floor(float(Wp) * sf))
        inv = 1.0 / sf
        Hf, Wf = int(math.floor(float(hs) * inv)), int(math.floor(float(ws) * inv))
        return Hp, Wp, hs, ws, Hf, Wf

    def forward(self, frames0, frames1, sf, embt, out):
        """IRFNet.forward(img0, img1, scale_factor=sf, timestep=embt) for N pairs; frames: lists of [H,W,C>=3] fp32 device
        tensors; out [N,H,W,3] device tensor (already clamped to [0,1] by the network)."""
        lib, st = self.lib, _lib.stream_ptr()
        N = len(frames0)
        H, W, Cc = frames0[0].shape
        sf = float(sf)
        Hp, Wp, hs, ws, Hf, Wf = self.geometry(H, W, sf)
        if sf <= 0 or hs < 16 or ws < 16 or hs % 16 or ws % 16:
            raise RuntimeError(f"IFRNet: working resolution {hs}x{ws} (padded {Hp}x{Wp} frame x {sf}) is not a multiple of 16; "
                               f"the reference fails here too (torch.cat of mismatched pyramid levels)")
        if Hf < H or Wf < W:
            raise RuntimeError(f"IFRNet: resizing {hs}x{ws} back by 1/{sf} gives {Hf}x{Wf}, smaller than the {H}x{W} frame; "
                               f"the reference fails here too (frame does not fit the output tensor)")
        assert out.is_cuda and out.is_contiguous() and tuple(out.shape) == (N, H, W, 3)
        img = self._t("img", 2 * N, Hp, Wp, 4)
        for n in range(N):
            f0, f1 = frames0[n], frames1[n]
            assert f0.is_cuda and f0.is_contiguous() and f1.is_contiguous() and f0.shape == f1.shape == (H, W, Cc) and Cc >= 3
            _lib.check(lib.vfi_ifrnet_prep(f0.data_ptr(), f1.data_ptr(), Cc, H, W, _p(img[n]), _p(img[N + n]), Hp, Wp, st), "vfi_ifrnet_prep")
        cm, mean = self._t("cm", 2 * N, 4), self._t("mean", max(N, 1))
        rm = self._t("rowmean", 2 * N, Hp, 4)     # row means first (one block per row), then the mean of the rows
        _lib.check(lib.vfi_pool_mean(_p(img), 4, _p(rm), 4, 2 * N, Hp, Wp, 4, 1, st), "vfi_pool_mean")
        _lib.check(lib.vfi_pool_mean(_p(rm), 4, _p(cm), 4, 2 * N, Hp, 1, 4, 0, st), "vfi_pool_mean")
        _lib.check(lib.vfi_ifrnet_center(_p(img), _p(cm), _p(mean), N, Hp * Wp, st), "vfi_ifrnet_center")
        r = self._t("r", 2 * N, hs, ws, 8)
        self._resize(img, 0, r, 0, 3, 1.0 / sf, 1.0)
        # ---- encoder, both images as one batch of 2N
        feats, x = [], r
        for lvl, c in enumerate(self.widths):
            h, w = hs >> (lvl + 1), ws >> (lvl + 1)
            a, f = self._t(f"enc_a{lvl}", 2 * N, h, w, _cs(c)), self._t(f"enc_f{lvl}", 2 * N, h, w, _cs(c))
            first, second = self.enc[lvl]
            if first is None:
                wt, bs, sl = self.head
                _lib.check(lib.vfi_conv7x7s2_prelu(_p(x), x.shape[-1], _p(wt), _p(bs), _p(sl), c, _p(a), a.shape[-1], 2 * N, x.shape[1],
                                                   x.shape[2], st), "vfi_conv7x7s2_prelu")
            else:
                first(x, 0, a, 0)
            second(a, 0, f, 0)
            feats.append(f)
            x = f
        # ---- decoder 4: cat(f0_4, f1_4, embt)
        c4 = self.widths[3]
        h, w = hs >> 4, ws >> 4
        din = self._t("d4in", N, h, w, _cs(2 * c4 + 1))
        self._ax(feats[3][:N], 0, None, 0, din, 0, c4)
        self._ax(feats[3][N:], 0, None, 0, din, c4, c4)
        vals = (C.c_float * N)(*([float(embt)] * N))
        _lib.check(lib.vfi_fill_items(_p(din, 2 * c4), din.shape[-1], 1, N, h * w, vals, st), "vfi_fill_items")
        flow_src = None     # tensor whose channels 0:4 hold the previous level's up_flow
        for d, lvl in ((4, 3), (3, 2), (2, 1), (1, 0)):
            if d < 4:
                # din already holds decoder d+1's output in channels [0, 4+f): flow0, flow1, ft_
                f = self.widths[lvl]
                if flow_src is not None:   # up_flow = out[:, 0:4] + 2.0 * resize(previous up_flow, 2.0)
                    tf = self._t("tf", N, din.shape[1], din.shape[2], 4)
                    self._up2(flow_src, tf, 4)
                    self._ax(din, 0, tf, 0, din, 0, 4)
                self._warp(feats[lvl][:N], f, din, 0, din, 4 + f)
                self._warp(feats[lvl][N:], f, din, 2, din, 4 + 2 * f)
            if d > 1:
                fn = self.widths[lvl - 1]
                target = self._t(f"d{d - 1}in", N, din.shape[1] * 2, din.shape[2] * 2, _cs(4 + 3 * fn))
            else:
                target = self._t("out1", N, hs, ws, 8)
            self._decoder(d, din, target)
            flow_src = din if d < 4 else None
            din = target
        out1 = din
        # up_flow_1 = out1[:, 0:4] + 2.0 * resize(up_flow_2, 2.0); up_mask_1 = sigmoid(out1[:, 4:5])
        tf = self._t("tf", N, hs, ws, 4)
        self._up2(flow_src, tf, 4)
        self._ax(out1, 0, tf, 0, out1, 0, 4)
        _lib.check(lib.vfi_sigmoid(_p(out1, 4), 8, 1, N * hs * ws, st), "vfi_sigmoid")
        fin = self._t("fin", N, Hf, Wf, 8)
        inv = 1.0 / sf
        self._resize(out1, 0, fin, 0, 4, 1.0 / inv, inv)       # resize(up_flow, 1/sf) * (1/sf)
        self._resize(out1, 4, fin, 4, 4, 1.0 / inv, 1.0)       # mask, residual
        _lib.check(lib.vfi_ifrnet_output(_p(img[:N]), _p(img[N:]), _p(fin), _p(mean), out.data_ptr(), N, Hp, Wp, Hf, Wf, H, W, st),
                   "vfi_ifrnet_output")
        return out

    # (pair, timestep) interface of m2m.run_plan: the node's timestep is forward()'s scale_factor, self.embt its timestep
    def prepare(self, frame0, frame1):
        self._pair = (frame0, frame1)

    def render(self, t, out):
        f0, f1 = self._pair
        self.forward([f0], [f1], t, self.embt, out.view((1,) + tuple(out.shape)))
        return out


def _load_state_dict(path):
    sd = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(sd, dict) and "state_dict" in sd and not any(k.startswith("encoder.") for k in sd):
        sd = sd["state_dict"]
    return sd


class IFRNet_VFI:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "ckpt_name": (CKPT_NAMES,),
                "frames": ("IMAGE",),
                "clear_cache_after_n_frames": ("INT", {"default": 10, "min": 1, "max": 1000}),
                "multiplier": ("INT", {"default": 2, "min": 2, "max": 1000}),
                "scale_factor": ([0.25, 0.5, 1.0, 2.0, 4.0], {"default": 1.0}),
            },
            "optional": {"optional_interpolation_states": ("INTERPOLATION_STATES",)},
        }

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "vfi"
    CATEGORY = "ComfyUI-Frame-Interpolation/VFI"

    def vfi(self, ckpt_name: typing.AnyStr, frames: torch.Tensor, clear_cache_after_n_frames: typing.SupportsInt = 1,
            multiplier: typing.SupportsInt = 2, scale_factor: typing.SupportsFloat = 1.0,
            optional_interpolation_states: InterpolationStateList = None, **kwargs):
        from .m2m import run_plan

        assert len(frames) >= 2, f"VFI model IFRNet requires at least 2 frames to work with, only found {frames.shape[0]}."
        model_path = load_file_from_github_release(MODEL_TYPE, ckpt_name)
        kind = kind_of(ckpt_name)
        # (the reference rebuilds the model on every call, ifrnet/__init__.py:42-45; see ckpt.cached_engine)
        def build():
            sd = _load_state_dict(model_path)
            return lane_set("ifrnet", lambda: IFRNetEngine(sd, kind))
        engine, cached = cached_engine(MODEL_TYPE + kind, model_path, build)
        try:
            begin_call(engine, tuple(frames.shape[1:3]) + (float(scale_factor), multiplier if isinstance(multiplier, int) else -1))
            embt = float(scale_factor)           # positional mis-binding of the reference's call, see the module docstring
            configure_lanes(engine, lambda e: setattr(e, "embt", embt))
            plan, tasks = generic_output_plan(len(frames), multiplier, optional_interpolation_states)
            return (run_plan(engine, frames, plan, tasks, name="IFRNet VFI"),)
        finally:
            if cached:
                torch.cuda.synchronize(engine.device)
            end_call(engine, cached)      # (the scratch tensors stay for the next call of this frame shape: ckpt.KEEP_WORKSPACE_BYTES)
