"""FILM VFI node — host-side mirror of the reference's ``FILM_VFI`` over the HIP library.

Same node shape and scheduling as vfi_models/film/__init__.py:12-113 (greedy bisection per pair, skipped pairs are
dropped, outputs re-used as inputs after ``clamp(0,1)``); the interpolator itself (vfi_models/film/film_arch.py
``Interpolator.debug_forward``, the source mirror of the TorchScript artifact the reference loads) is executed
by the C-side object vfi_film_* (csrc/film_net.hip) as a sequence of the library's generic ops: every torch.nn.functional call
of the reference maps to one entry point

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
from .lanes import lane_set, lanes_of, tell_lone_pair
from .dist import all_gather_frames, world
from .film_spec import check_state_dict, film_shapes
from .schedule import InterpolationStateList, shard_tasks

MODEL_TYPE = "film"


class FilmEngine:
    """Device-resident FILM interpolator (one frame pair per call, like the reference node): the C-side object
    vfi_film_create / vfi_film_forward / vfi_film_destroy (csrc/film_net.hip) — weights packed once, workspace owned by the
    library, the whole launch sequence of a pair issued by one call."""

    def __init__(self, state_dict, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("FILM VFI (HIP): no GPU visible; this node has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _lib.check(self.lib.vfi_init(self.device.index or 0), "vfi_init")
        check_state_dict(state_dict)
        keys = list(film_shapes().keys())
        tensors = [state_dict[k].detach().to("cpu", torch.float32).contiguous() for k in keys]
        ptrs = (C.c_void_p * len(keys))(*[t.data_ptr() for t in tensors])
        numels = (C.c_int64 * len(keys))(*[t.numel() for t in tensors])
        self.handle = self.lib.vfi_film_create(ptrs, numels, len(keys))
        if not self.handle:
            raise RuntimeError("vfi_film_create failed: " + _lib.last_error())

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vfi_film_destroy(self.handle)
            self.handle = None

    def release_workspace(self):
        """Drop the activations (15 GB at 1080p); the packed weights stay on the device."""
        _lib.check(self.lib.vfi_film_release_workspace(self.handle), "vfi_film_release_workspace")

    def two_streams(self, on):
        """vfi_film_forward forks half of the network onto the object's side stream (default) / stays on the caller's stream (what the
        node wants once several pairs are in flight on lanes of their own).  Bit-identical frames either way."""
        return bool(self.lib.vfi_film_two_streams(self.handle, int(bool(on))))

    lone_pair = two_streams      # lanes.tell_lone_pair

    def debug_flow(self, d, level, h, w):
        """test tap: flow pyramid level of the last forward, direction d (0 forward, 1 backward) -> [h,w,2] host tensor"""
        buf = torch.empty(h * w * 2, dtype=torch.float32)
        n = _lib.test_tap("vfi_film_debug_read_flow")(self.handle, d, level, buf.data_ptr(), buf.numel())
        if n != buf.numel():
            raise RuntimeError("vfi_film_debug_read_flow: " + _lib.last_error())
        return buf.view(h, w, 2)

    def forward(self, x0, x1, clamp=False):
        """x0, x1: [H,W,C>=3] fp32 device tensors -> [H,W,3] device tensor (Interpolator.forward, time = 0.5)."""
        H, W = x0.shape[:2]
        for x in (x0, x1):
            assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape[:2]) == (H, W) and x.shape[2] == x0.shape[2]
        out = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.vfi_film_forward(self.handle, x0.data_ptr(), x1.data_ptr(), x0.shape[2], H, W, out.data_ptr(), int(bool(clamp)),
                                             _lib.stream_ptr()), "vfi_film_forward")
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
        def build():
            sd = _load_state_dict(model_path)
            return lane_set("film", lambda: FilmEngine(sd))
        engine, cached = cached_engine(MODEL_TYPE, model_path, build)
        try:
            frames = frames[..., :3]
            n = len(frames)
            if type(multiplier) == int:
                multipliers = [multiplier] * n
            else:
                multipliers = list(map(int, multiplier))
                multipliers += [2] * (n - len(multipliers) - 1)
            # inference(..., inter_frames = m - 1) (film/__init__.py:12-41): m in {-1, 0, 1} runs no iteration and the pair contributes
            # frame_i alone; m <= -2 fails in torch.linspace(0, 1, m + 1)
            kept = [i for i in range(n - 1)
                    if not (optional_interpolation_states is not None and optional_interpolation_states.is_frame_skipped(i))]
            if any(multipliers[i] <= -2 for i in kept):
                raise RuntimeError(f"FILM: multiplier {min(multipliers[i] for i in kept)} — the reference fails in torch.linspace for multipliers <= -2")
            multipliers = [max(int(m), 1) for m in multipliers]
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
            # pair lanes (lanes.py): kept pair j runs on lane j % n_lanes = its own engine on its own stream (the bisection inside a
            # pair stays sequential on that stream); `main` only carries the bookkeeping events
            main = torch.cuda.current_stream(dev)
            if hasattr(engine, "apart_from"):      # (a LaneSet) its streams stay clear of the copy streams' hardware queues where there are enough
                from .hostpipe import _stream
                engine.apart_from = [_stream(dev, "down"), _stream(dev, "up"), main]
            lane, n_lanes = lanes_of(engine, len(mine))
            tell_lone_pair(engine, n_lanes)      # the forward's own two-stream fork is for a lone pair
            up = Uploader(frames, order, dev, main, depth=min(max(4, n_lanes + 2), len(order)) or 1)
            keep, pos, released, pending = [], 0, 0, []
            try:
                for j in range(lo, hi):
                    i = pairs[j]
                    eng, st = lane((j - lo) % n_lanes)
                    res = {0: up.get(item_of[i], st), multipliers[i]: up.get(item_of[i + 1], st)}
                    with torch.cuda.stream(st):
                        for (l, r, new) in film_schedule(multipliers[i] - 1):
                            res[new] = eng.forward(res[l], res[r], clamp=True)
                        for n_k, k in enumerate(sorted(res)[:-1]):
                            if ws > 1:
                                local[pos] = res[k]      # res[0] is the uploaded original: bit-exact round trip
                            elif k == 0:
                                wr.put_host(row0[j] + n_k, frames[i])
                            else:
                                wr.put_dev(row0[j] + n_k, res[k], st)
                                keep.append(res[k])      # alive until the copy-back has read it
                            pos += 1
                        if n_lanes > 1:
                            done = torch.cuda.Event()
                            done.record(st)
                            pending.append(done)
                    if released < item_of[i + 1]:        # frames before i+1 are not needed again (pairs ascend)
                        for ev in pending:               # ... once every lane that read them is through
                            main.wait_event(ev)
                        pending = []
                        while released < item_of[i + 1]:
                            up.release(released)
                            released += 1
            finally:
                for ev in pending:
                    main.wait_event(ev)
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
