"""FILM VFI node — host-side mirror of the reference's ``FILM_VFI`` over the HIP library.

Same node shape and scheduling as vfi_models/film/__init__.py:12-113 (greedy bisection per pair, skipped pairs are
dropped, outputs re-used as inputs after ``clamp(0,1)``); the interpolator itself (vfi_models/film/film_arch.py
``Interpolator.debug_forward``, the source mirror of the TorchScript artifact the reference loads) is executed
as a sequence of C-ABI calls: every torch.nn.functional call of the reference maps to one library entry point

    F.conv2d(padding='same') (+LeakyReLU)  -> vfi_conv_forward     (fp32 MFMA implicit GEMM)
    F.avg_pool2d(2,2)                       -> vfi_avgpool2
    F.grid_sample via film_arch.warp        -> vfi_warp_film
    F.interpolate(bilinear / nearest)       -> vfi_resize_bilinear / vfi_upsample_nearest
    +, *scalar                              -> vfi_axpby

and every ``torch.cat`` along channels disappears: producers write into channel windows of pre-allocated NHWC
tensors (physical channel positions are translated once, at weight-pack time, through ``chan_map``).
torch only allocates device buffers and moves frames (plumbing).
"""
import bisect
import ctypes as C
import typing

import numpy as np
import torch

from . import _lib
from .ckpt import cached_engine, load_file_from_github_release
from .dist import all_gather_frames, world
from .film_spec import FLOW_FILTERS, check_state_dict, feat_channels
from .schedule import InterpolationStateList, shard_tasks

MODEL_TYPE = "film"
PYR, FUS = 7, 5


def _r8(n):
    return (n + 7) // 8 * 8


class _Conv:
    def __init__(self, lib, sd, key, chan_map=None, cin_phys=None):
        w = sd[key + ".weight"].detach().to("cpu", torch.float32).contiguous()
        b = sd[key + ".bias"].detach().to("cpu", torch.float32).contiguous()
        cout, cin, kh, kw = w.shape
        self.cout, self.cin_phys = cout, cin_phys or _r8(cin)
        cm = None
        if chan_map is not None:
            cm = (C.c_int * cin)(*chan_map)
        self.lib = lib
        self.h = lib.vfi_conv_create(w.data_ptr(), b.data_ptr(), cout, cin, kh, kw, cm, self.cin_phys)
        if not self.h:
            raise RuntimeError(f"vfi_conv_create({key}) failed: " + _lib.last_error())

    def __call__(self, src, src_off, dst, dst_off, H, W, act=1):
        """src/dst: contiguous [H,W,C] device tensors; *_off = first channel of the window."""
        _lib.check(self.lib.vfi_conv_forward(self.h, src.data_ptr() + 4 * src_off, src.shape[-1], dst.data_ptr() + 4 * dst_off,
                                             dst.shape[-1], 1, H, W, act, 0.2, _lib.stream_ptr()), "vfi_conv_forward")

    def close(self):
        if self.h:
            self.lib.vfi_conv_destroy(self.h)
            self.h = None


def _aligned_map(F):
    """reference channel order of one aligned-pyramid level  [imgA 3, featA F, imgB 3, featB F, bflow 2, fflow 2]
    -> physical order  [imgA 3, 0, featA F | imgB 3, 0, featB F | bflow 2, fflow 2 | zero pad to x8]."""
    m = []
    for half in range(2):
        base = half * (4 + F)
        m += [base + c for c in range(3)]
        m += [base + 4 + c for c in range(F)]
    m += [2 * (4 + F) + c for c in range(4)]
    return m


class FilmEngine:
    """Device-resident FILM interpolator (one frame pair per call, like the reference node)."""

    def __init__(self, state_dict, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("FILM VFI (HIP): no GPU visible; this node has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _lib.check(self.lib.vfi_init(self.device.index or 0), "vfi_init")
        check_state_dict(state_dict)
        sd, lib = state_dict, self.lib
        self.ext = [[_Conv(lib, sd, f"extract.extract_sublevels.convs.{i}.{j}.0") for j in range(2)] for i in range(4)]

        def estimator(prefix):
            return [_Conv(lib, sd, f"{prefix}._convs.{i}.0") for i in range(4)] + [_Conv(lib, sd, f"{prefix}._convs.4")]

        # flow predictor per level: 0,1,2 specialised (state_dict index k = 2 - level), >=3 shared
        self.pred = {2: estimator("predict_flow._predictors.0"), 1: estimator("predict_flow._predictors.1"),
                     0: estimator("predict_flow._predictors.2"), 3: estimator("predict_flow._predictor")}
        self.cal = [_r8(2 * (4 + feat_channels(l)) + 4) for l in range(FUS)]
        self.fuse = []
        for k in range(4):
            lvl = 3 - k
            nf = 64 << min(lvl, 3)
            if k == 0:
                c0 = _Conv(lib, sd, "fuse.convs.0.0", _aligned_map(feat_channels(4)), self.cal[4])
            else:
                c0 = _Conv(lib, sd, f"fuse.convs.{k}.0")
            skip = _aligned_map(feat_channels(lvl))
            c1 = _Conv(lib, sd, f"fuse.convs.{k}.1.0", skip + [self.cal[lvl] + c for c in range(nf)], self.cal[lvl] + nf)
            c2 = _Conv(lib, sd, f"fuse.convs.{k}.2.0")
            self.fuse.append((c0, c1, c2, nf))
        self.out_conv = _Conv(lib, sd, "fuse.output_conv")
        self.shape = None

    def close(self):
        for grp in [c for st in self.ext for c in st] + [c for p in self.pred.values() for c in p] + \
                   [c for f in self.fuse for c in f[:3]] + [self.out_conv]:
            grp.close()

    # ------------------------------------------------------------------------------------------------
    def _z(self, h, w, c):
        return torch.zeros((h, w, c), dtype=torch.float32, device=self.device)

    def _alloc(self, H, W):
        if self.shape == (H, W):
            return
        assert H >= 64 and W >= 64, "FILM needs 7 pyramid levels (>= 64 px per side)"
        self.hw = [(H, W)]
        for _ in range(PYR - 1):
            self.hw.append((self.hw[-1][0] // 2, self.hw[-1][1] // 2))
        hw = self.hw
        self.img = [[self._z(*hw[l], 8) for l in range(PYR)] for _ in range(2)]
        self.tw = [[self._z(*hw[l], 4 + feat_channels(l)) for l in range(PYR)] for _ in range(2)]
        self.pair = [self._z(*hw[l], 2 * feat_channels(l)) for l in range(PYR)]
        self.flow = [[self._z(*hw[l], 2) for l in range(PYR)] for _ in range(2)]     # v per level, per direction
        self.vres = [self._z(*hw[l], 2) for l in range(PYR)]
        self.vup = [self._z(*hw[l], 2) for l in range(PYR)]
        self.al = [self._z(*hw[l], self.cal[l] + (64 << min(l, 3) if l < 4 else 0)) for l in range(FUS)]
        self.scratch = {}
        self.shape = (H, W)

    def release_workspace(self):
        """Drop the activations (15 GB at 1080p); the packed weights stay on the device."""
        self.img = self.tw = self.pair = self.flow = self.vres = self.vup = self.al = None
        self.scratch = {}
        self.shape = None

    def _tmp(self, name, h, w, c):
        key = (name, h, w, c)
        if key not in self.scratch:
            self.scratch[key] = self._z(h, w, c)
        return self.scratch[key]

    def _ax(self, a, a_off, b, b_off, out, out_off, h, w, c, alpha=1.0, beta=1.0):
        _lib.check(self.lib.vfi_axpby(a.data_ptr() + 4 * a_off, a.shape[-1], (b.data_ptr() + 4 * b_off) if b is not None else None,
                                      b.shape[-1] if b is not None else 0, out.data_ptr() + 4 * out_off, out.shape[-1], h * w, c,
                                      alpha, beta, _lib.stream_ptr()), "vfi_axpby")

    def _extract(self, k):
        """FeatureExtractor on image pyramid k: writes the cascaded feature pyramid into tw[k][*][..., 4:]."""
        hw, lib, st = self.hw, self.lib, _lib.stream_ptr
        for i in range(PYR):
            depth = min(PYR - i, 4)
            src, src_off = self.img[k][i], 0
            for j in range(depth):
                lvl = i + j
                h, w = hw[lvl]
                c = 64 << j
                mid = self._tmp("ext_mid", h, w, c)
                self.ext[j][0](src, src_off, mid, 0, h, w)
                slot_off = 4 + sum(64 << q for q in range(j))     # position of sub[.][j] inside level lvl's features
                self.ext[j][1](mid, 0, self.tw[k][lvl], slot_off, h, w)
                if j < depth - 1:
                    h2, w2 = hw[lvl + 1]
                    pooled = self._tmp("ext_pool", h2, w2, c)
                    _lib.check(lib.vfi_avgpool2(self.tw[k][lvl].data_ptr() + 4 * slot_off, self.tw[k][lvl].shape[-1],
                                                pooled.data_ptr(), c, 1, h, w, c, st()), "vfi_avgpool2")
                    src, src_off = pooled, 0

    def _predict(self, a, b, d):
        """PyramidFlowEstimator(feature pyramid a, feature pyramid b) -> self.flow[d][l] = synthesised flow per level."""
        hw, lib, st = self.hw, self.lib, _lib.stream_ptr
        for l in range(PYR - 1, -1, -1):
            h, w = hw[l]
            F = feat_channels(l)
            pair = self.pair[l]
            self._ax(self.tw[a][l], 4, None, 0, pair, 0, h, w, F, 1.0, 0.0)
            if l == PYR - 1:
                self._ax(self.tw[b][l], 4, None, 0, pair, F, h, w, F, 1.0, 0.0)
            else:
                h1, w1 = hw[l + 1]
                _lib.check(lib.vfi_resize_bilinear(self.flow[d][l + 1].data_ptr(), 2, self.vup[l].data_ptr(), 2, 1, h1, w1, h, w, 2,
                                                   2.0, st()), "vfi_resize_bilinear")
                _lib.check(lib.vfi_warp_film(self.tw[b][l].data_ptr() + 16, self.tw[b][l].shape[-1], self.vup[l].data_ptr(), 2, 1.0,
                                             pair.data_ptr() + 4 * F, pair.shape[-1], 1, h, w, F, st()), "vfi_warp_film")
            convs = self.pred[min(l, 3)]
            nf = FLOW_FILTERS[min(l, 3)]
            t0, t1 = self._tmp("fe0", h, w, nf), self._tmp("fe1", h, w, nf)
            convs[0](pair, 0, t0, 0, h, w)
            convs[1](t0, 0, t1, 0, h, w)
            convs[2](t1, 0, t0, 0, h, w)
            t2 = self._tmp("fe2", h, w, _r8(nf // 2))
            convs[3](t0, 0, t2, 0, h, w)
            if l == PYR - 1:
                convs[4](t2, 0, self.flow[d][l], 0, h, w, act=0)
            else:
                convs[4](t2, 0, self.vres[l], 0, h, w, act=0)
                self._ax(self.vres[l], 0, self.vup[l], 0, self.flow[d][l], 0, h, w, 2, 1.0, 1.0)   # v = v_residual + v

    def forward(self, x0, x1, clamp=False):
        """x0, x1: [H,W,C>=3] fp32 device tensors -> [H,W,3] device tensor (Interpolator.forward, time = 0.5)."""
        H, W = x0.shape[:2]
        self._alloc(H, W)
        hw, lib, st = self.hw, self.lib, _lib.stream_ptr
        for k, x in enumerate((x0, x1)):
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[:2] == (H, W)
            self._ax(x, 0, None, 0, self.img[k][0], 0, H, W, 3, 1.0, 0.0)
            for l in range(1, PYR):     # build_image_pyramid
                _lib.check(lib.vfi_avgpool2(self.img[k][l - 1].data_ptr(), 8, self.img[k][l].data_ptr(), 8, 1, *hw[l - 1], 4, st()),
                           "vfi_avgpool2")
            for l in range(PYR):        # image part of the to-warp pyramids
                self._ax(self.img[k][l], 0, None, 0, self.tw[k][l], 0, *hw[l], 4, 1.0, 0.0)
            self._extract(k)
        self._predict(0, 1, 0)   # forward  residual flow pyramid
        self._predict(1, 0, 1)   # backward residual flow pyramid
        # aligned pyramid: [warp(pyr0, 0.5*bwd) | warp(pyr1, 0.5*fwd) | 0.5*bwd | 0.5*fwd]
        for l in range(FUS):
            h, w = hw[l]
            F = feat_channels(l)
            al = self.al[l]
            for half, (src, fl) in enumerate(((self.tw[0][l], self.flow[1][l]), (self.tw[1][l], self.flow[0][l]))):
                _lib.check(lib.vfi_warp_film(src.data_ptr(), src.shape[-1], fl.data_ptr(), 2, 0.5, al.data_ptr() + 4 * half * (4 + F),
                                             al.shape[-1], 1, h, w, 4 + F, st()), "vfi_warp_film")
            self._ax(self.flow[1][l], 0, None, 0, al, 2 * (4 + F), h, w, 2, 0.5, 0.0)
            self._ax(self.flow[0][l], 0, None, 0, al, 2 * (4 + F) + 2, h, w, 2, 0.5, 0.0)
        # Fusion
        net, net_c, nh, nw = self.al[4], self.cal[4], *hw[4]
        for k, (c0, c1, c2, nf) in enumerate(self.fuse):
            lvl = 3 - k
            h, w = hw[lvl]
            up = self._tmp("fuse_up", h, w, net_c)
            _lib.check(lib.vfi_upsample_nearest(net.data_ptr(), net.shape[-1], up.data_ptr(), net_c, 1, nh, nw, h, w, net_c, st()),
                       "vfi_upsample_nearest")
            c0(up, 0, self.al[lvl], self.cal[lvl], h, w, act=0)
            t = self._tmp("fuse_t", h, w, nf)
            c1(self.al[lvl], 0, t, 0, h, w)
            o = self._tmp("fuse_o", h, w, nf)
            c2(t, 0, o, 0, h, w)
            net, net_c, nh, nw = o, nf, h, w
        out = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
        self.out_conv(net, 0, out, 0, H, W, act=2 if clamp else 0)
        return out


def film_schedule(inter_frames):
    """Greedy bisection order of one pair (vfi_models/film/__init__.py:17-40): list of (left, right, new) grid
    positions; the model is always asked for the midpoint (it ignores dt, film_arch.py:427-429)."""
    idxes = [0, inter_frames + 1]
    remains = list(range(1, inter_frames + 1))
    splits = torch.linspace(0, 1, inter_frames + 2)
    calls = []
    for _ in range(len(remains)):
        starts = splits[idxes[:-1]]
        ends = splits[idxes[1:]]
        distances = ((splits[None, remains] - starts[:, None]) / (ends[:, None] - starts[:, None]) - .5).abs()
        start_i, step = np.unravel_index(torch.argmin(distances).item(), distances.shape)
        new = remains[step]
        calls.append((idxes[start_i], idxes[start_i + 1], new))
        idxes.insert(bisect.bisect_left(idxes, new), new)
        del remains[step]
    return calls


def _load_state_dict(path):
    try:
        return torch.jit.load(path, map_location="cpu").state_dict()   # the reference's artifact (film/__init__.py:74)
    except Exception:
        sd = torch.load(path, map_location="cpu", weights_only=False)
        return sd.state_dict() if hasattr(sd, "state_dict") else sd


class FILM_VFI:
    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "ckpt_name": (["film_net_fp32.pt"],),
                "frames": ("IMAGE",),
                "clear_cache_after_n_frames": ("INT", {"default": 10, "min": 1, "max": 1000}),
                "multiplier": ("INT", {"default": 2, "min": 2, "max": 1000}),
            },
            "optional": {"optional_interpolation_states": ("INTERPOLATION_STATES",)},
        }

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "vfi"
    CATEGORY = "ComfyUI-Frame-Interpolation/VFI"

    def vfi(self, ckpt_name: typing.AnyStr, frames: torch.Tensor, clear_cache_after_n_frames=10,
            multiplier: typing.SupportsInt = 2, optional_interpolation_states: InterpolationStateList = None, **kwargs):
        model_path = load_file_from_github_release(MODEL_TYPE, ckpt_name)
        # (the reference re-loads the TorchScript file on every call, film/__init__.py:74; see ckpt.cached_engine)
        engine, cached = cached_engine(MODEL_TYPE, model_path, lambda: FilmEngine(_load_state_dict(model_path)))
        try:
            frames = frames[..., :3]
            n = len(frames)
            if type(multiplier) == int:
                multipliers = [multiplier] * n
            else:
                multipliers = list(map(int, multiplier))
                multipliers += [2] * (n - len(multipliers) - 1)
            dev = engine.device
            H, W = frames.shape[1:3]
            # pairs are independent (the bisection inside a pair is sequential): block-partition the kept pairs over
            # ranks, all-gather the per-rank frame blocks
            pairs = [i for i in range(n - 1)
                     if not (optional_interpolation_states is not None and optional_interpolation_states.is_frame_skipped(i))]
            rank, ws = world()
            lo, hi = shard_tasks(pairs, rank, ws)
            per_pair = [multipliers[i] for i in pairs]           # output frames contributed by each kept pair
            counts = [sum(per_pair[slice(*shard_tasks(pairs, r, ws))]) for r in range(ws)]
            # host side (hostpipe.py): new frames and pass-through frames land in their rows of the output tensor in the
            # background while the next pair computes
            from .hostpipe import OutputWriter, Uploader
            wr = OutputWriter(sum(per_pair) + 1, H, W, dev)
            row0 = [0]
            for m in per_pair:
                row0.append(row0[-1] + m)          # first output row of each kept pair
            local = torch.empty((counts[rank], H, W, 3), dtype=torch.float32, device=dev) if ws > 1 else None
            mine = pairs[lo:hi]
            order = sorted({f for i in mine for f in (i, i + 1)})       # every needed frame is uploaded once, ahead of use
            item_of = {f: k for k, f in enumerate(order)}
            up = Uploader(frames, order, dev, torch.cuda.current_stream(dev), depth=min(4, len(order)) or 1)
            keep, pos, held, released = [], 0, {}, 0
            try:
                for j in range(lo, hi):
                    i = pairs[j]
                    for f in (i, i + 1):
                        if f not in held:
                            held[f] = up.get(item_of[f])
                    res = {0: held[i], multipliers[i]: held[i + 1]}
                    for (l, r, new) in film_schedule(multipliers[i] - 1):
                        res[new] = engine.forward(res[l], res[r], clamp=True)
                    for n_k, k in enumerate(sorted(res)[:-1]):
                        if ws > 1:
                            local[pos] = res[k]      # res[0] is the uploaded original: bit-exact round trip
                        elif k == 0:
                            wr.put_host(row0[j] + n_k, frames[i])
                        else:
                            wr.put_dev(row0[j] + n_k, res[k])
                            keep.append(res[k])      # alive until the copy-back has read it
                        pos += 1
                    while released < item_of[i + 1]:     # frames before i+1 are not needed again (pairs ascend)
                        up.release(released)
                        held.pop(order[released], None)
                        released += 1
            finally:
                up.close()
            if ws > 1:
                allf = all_gather_frames(local, counts)
                for k in range(allf.shape[0]):
                    wr.put_dev(k, allf[k])
            wr.put_host(sum(per_pair), frames[-1])
            return (wr.finish(),)
        finally:
            if cached:
                torch.cuda.synchronize(engine.device)
                engine.release_workspace()
            else:
                engine.close()
