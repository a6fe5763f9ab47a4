"""M2M VFI node — host-side mirror of the reference's ``M2M_VFI`` over the HIP library.

Node shape and frame loop follow vfi_models/m2m/__init__.py:14-60 + vfi_utils.generic_frame_loop
(vfi_utils.py:149-389, timestep mode): frame_i, its multiplier-1 middle frames, ..., last frame; skipped pairs keep
their first frame; no clamp.  The model (vfi_models/m2m/M2M_arch.py ``M2M_PWC.forward`` :894-1037) is executed as
a sequence of the library's generic ops, issued by the C-side object vfi_m2m_* (csrc/m2m_object.hip):

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
from .ckpt import cached_engine, end_call, load_file_from_github_release
from .lanes import LaneSet, lane_set
from .dist import all_gather_frames, world
from .m2m_spec import check_state_dict, m2m_shapes
from .schedule import InterpolationStateList, generic_output_plan, shard_tasks

MODEL_TYPE = "m2m"
RATIO = 4        # M2M_PWC.forward default ratio (the node never overrides it, vfi_models/m2m/__init__.py:51-55)


class M2MEngine:
    """Device-resident M2M interpolator: ``prepare(frame0, frame1)`` once per pair, ``render(t)`` per timestep — the C-side
    object vfi_m2m_create / vfi_m2m_prepare / vfi_m2m_render / vfi_m2m_destroy (csrc/m2m_object.hip): weights packed once,
    workspace owned by the library, the ~110 launches of a pair issued by one call."""

    def __init__(self, state_dict, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("M2M VFI (HIP): no GPU visible; this node has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _lib.check(self.lib.vfi_init(self.device.index or 0), "vfi_init")
        check_state_dict(state_dict)
        keys = list(m2m_shapes().keys())
        tensors = [state_dict[k].detach().to("cpu", torch.float32).contiguous() for k in keys]
        ptrs = (C.c_void_p * len(keys))(*[t.data_ptr() for t in tensors])
        numels = (C.c_int64 * len(keys))(*[t.numel() for t in tensors])
        self.handle = self.lib.vfi_m2m_create(ptrs, numels, len(keys))
        if not self.handle:
            raise RuntimeError("vfi_m2m_create failed: " + _lib.last_error())
        self.hw = None

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vfi_m2m_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def release_workspace(self):
        """Drop the activations; the packed weights stay on the device."""
        _lib.check(self.lib.vfi_m2m_release_workspace(self.handle), "vfi_m2m_release_workspace")
        self.hw = None

    def workspace_bytes(self):
        return int(self.lib.vfi_m2m_workspace_bytes(self.handle)) if getattr(self, "handle", None) else 0

    def prepare(self, frame0, frame1):
        """frame0/frame1: [H,W,C>=3] fp32 device tensors.  Runs everything that does not depend on the timestep (the frames are
        consumed by the first kernel of the call; later kernels read the library's own copies)."""
        H, W, Cc = frame0.shape
        assert frame1.shape == frame0.shape and Cc >= 3 and frame0.is_contiguous() and frame1.is_contiguous()
        assert frame0.is_cuda and frame0.dtype == torch.float32
        _lib.check(self.lib.vfi_m2m_prepare(self.handle, frame0.data_ptr(), frame1.data_ptr(), Cc, H, W, _lib.stream_ptr()), "vfi_m2m_prepare")
        self.hw = (H, W)

    def render(self, t, out=None):
        """One middle frame at time t for the prepared pair -> [H,W,3] device tensor (not clamped, like the reference)."""
        assert self.hw is not None, "prepare() first"
        H, W = self.hw
        if out is None:
            out = torch.empty((H, W, 3), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.vfi_m2m_render(self.handle, float(t), out.data_ptr(), _lib.stream_ptr()), "vfi_m2m_render")
        return out

    def forward(self, frame0, frame1, t):
        self.prepare(frame0, frame1)
        return self.render(t)

    # -- test taps (include/vfi_hip_test.h) ------------------------------------------------------------------------------
    def _debug(self, what, shape):
        buf = torch.empty(shape, dtype=torch.float32)
        n = _lib.test_tap("vfi_m2m_debug_read")(self.handle, what, buf.data_ptr(), buf.numel())
        if n != buf.numel():
            raise RuntimeError("vfi_m2m_debug_read: " + _lib.last_error())
        return buf

    def _padded(self):
        m = RATIO * 16
        return (self.hw[0] + m - 1) // m * m, (self.hw[1] + m - 1) // m * m

    @property
    def flow(self):
        Hp, Wp = self._padded()
        return [self._debug(0, (2, Hp // 4, Wp // 4, 2))]

    @property
    def d0(self):
        return self._debug(1, (2,) + self._padded() + (8,))

    @property
    def r(self):
        return self._debug(2, (2,) + self._padded() + (12,))


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
        engine = engine.engines[0] if isinstance(engine, LaneSet) else engine
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
    from .lanes import lanes_of, tell_lone_pair
    main = torch.cuda.current_stream(dev)
    wr = OutputWriter(len(plan), H, W, dev)
    new_row = {}
    for i, (kind, idx) in enumerate(plan):
        if kind == "src":
            wr.put_host(i, frames[idx])
        else:
            new_row[idx] = i
    mine = tasks[lo:hi]
    # pair lanes (lanes.py): pair j runs on lane j % n_lanes = its own engine on its own stream; `main` only carries the bookkeeping
    # events (a frame's staging slot is released on main after main has waited for every lane that read it)
    if isinstance(engine, LaneSet):      # the lanes' streams stay clear of the copy streams' hardware queues where there are enough of them
        from .hostpipe import _stream
        engine.apart_from = [_stream(dev, "down"), _stream(dev, "up"), main]
    lane, n_lanes = lanes_of(engine, len(mine))
    tell_lone_pair(engine, n_lanes)
    order = sorted({f for pair, _ in mine for f in (pair, pair + 1)})
    up = Uploader(frames, order, dev, main, depth=min(max(4, n_lanes + 2), len(order)) or 1)
    item_of = {f: i for i, f in enumerate(order)}
    local = torch.empty((counts[rank], H, W, 3), dtype=torch.float32, device=dev)
    first_new = sum(counts[:rank])
    pending = []          # completion events of lanes main has not waited for yet
    try:
        pos, released = 0, 0
        for j, (pair, ts) in enumerate(mine):
            eng, st = lane(j % n_lanes)
            f0, f1 = up.get(item_of[pair], st), up.get(item_of[pair + 1], st)
            with torch.cuda.stream(st):
                eng.prepare(f0, f1)
                for t in ts:
                    eng.render(t, local[pos])
                    if ws == 1:
                        wr.put_dev(new_row[first_new + pos], local[pos], st)
                    pos += 1
                if n_lanes > 1:
                    done = torch.cuda.Event()
                    done.record(st)
                    pending.append(done)
            # Release only after the LAST render of the pair: some engines' prepare() keeps references to the ring-slot
            # tensors and render() re-reads them (IFRNet, IFUNet), so the `consumed` event must follow those reads.
            if released < item_of[pair + 1]:          # frames before pair+1 are never needed again (tasks ascend)
                for ev in pending:
                    main.wait_event(ev)
                pending = []
                while released < item_of[pair + 1]:
                    up.release(released)
                    released += 1
        for ev in pending:
            main.wait_event(ev)
        pending = []
        if ws > 1:
            new = all_gather_frames(local, counts)
            for k in range(new.shape[0]):
                wr.put_dev(new_row[k], new[k])
    finally:
        for ev in pending:      # (an error path: the staging rings go back with a `busy` event recorded on main)
            main.wait_event(ev)
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
        def build():
            sd = _load_state_dict(model_path)
            return lane_set("m2m", lambda: M2MEngine(sd))
        engine, cached = cached_engine(MODEL_TYPE, model_path, build)
        try:
            plan, tasks = generic_output_plan(len(frames), multiplier, optional_interpolation_states)
            return (run_plan(engine, frames, plan, tasks),)
        finally:
            if cached:
                torch.cuda.synchronize(engine.device)
            end_call(engine, cached)      # (0.3 GB per lane at 1080p: stays for the next clip, ckpt.KEEP_WORKSPACE_BYTES)
