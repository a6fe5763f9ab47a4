"""RIFE VFI node — host-side mirror of the reference's ``RIFE_VFI`` over the HIP library.

Same class shape, widgets, call signature, output ordering and dtype as
vfi_models/rife/__init__.py:34-239 of the reference; the per-task hot loop
(``IFNet.forward`` + clamp) runs in libvfi_hip.so.  torch is used for what the prompt calls
plumbing: host<->device copies, the stream, and (multi-GPU) torch.distributed.

Differences that do not change results:
  * ``encode`` is evaluated once per input frame and cached on the device, not once per task;
  * ``fast_mode`` / ``ensemble`` are accepted and ignored — for arch 4.7 the reference's own call
    binds them to parameters that are inert (positional mis-binding, rife/__init__.py:200-207);
  * ``torch_compile`` is ignored (kernels are ahead-of-time compiled); ``dtype`` float16 / bfloat16
    rounds the clip to that dtype on the way in and returns the IMAGE tensor in it, like the reference,
    but computes in float32 (the parity contract is fp32, |d| <= 1e-3).
"""
import contextlib
import ctypes as C
import os
import threading
import typing
import warnings

import torch
from packaging import version

from . import _lib
from .ckpt import load_file_from_github_release
from .dist import all_gather_frames, world
from .rife_spec import ARCH_CODE, CKPT_NAME_VER_DICT, SUPPORTED_ARCH, check_state_dict, rife_shapes
from .schedule import InterpolationStateList, rife_output_plan, rife_task_list, shard_tasks

MODEL_TYPE = "rife"
DTYPE_OPTIONS = ["float32", "float16", "bfloat16"]
DTYPE_MAP = {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}   # rife/__init__.py:23-27
MAX_LIB_BATCH = 32  # kMaxTasks in csrc/rife_ops.h
# Tasks are independent and every task's arithmetic is the same whatever it is batched with (bit-exact since round 3: the conv
# tile variant is chosen from the image, not from the launch), so the node's `batch_size` widget (default 1,
# rife/__init__.py:68-71) only trades memory for speed.  288 GB of HBM make
# that trade moot: the node runs at least this many tasks per launch (8 below 4K, 4 from 4K up; 0 = honour the widget).
MIN_NODE_BATCH = int(os.environ.get("VFI_RIFE_MIN_BATCH", "8"))
PACK_AHEAD = 16            # frames packed on the device ahead of the launch that needs them (67 MB each at 1080p)
HOST_TIMELINE = False      # diagnostics (tools/node_e2e.py sets it): print the per-launch host timeline of run_tasks


def effective_batch(batch_size, H, W, n_tasks=None):
    """Tasks per launch.  With ``n_tasks`` (a host clip going through the upload / compute / download pipeline): at least four
    launches per call when the clip allows it, so that the first upload and the last download are a quarter of the clip
    instead of half of it — a 33-frame 1080p clip at the widget's 16 is two launches: 21 ms of upload before the GPU starts
    and 25 ms of download after it stops, around 58 ms of compute (fp32 frames); at 8 it is four, 11 + 60 + 12 ms."""
    floor_ = MIN_NODE_BATCH if H * W <= 2304 * 4096 // 2 else min(MIN_NODE_BATCH, 4)
    bs = max(1, min(MAX_LIB_BATCH, max(int(batch_size), floor_)))
    if n_tasks is not None and floor_ > 0:
        bs = min(bs, max(floor_, -(-int(n_tasks) // 4)))
    return bs


class RifeEngine:
    """Device-resident RIFE network (arch 4.7 = rife47/rife49, 4.17 = rife417, 4.26 = rife426) + frame cache behind the C ABI."""

    def __init__(self, state_dict, arch_ver="4.7", device=None):
        if arch_ver not in SUPPORTED_ARCH:
            raise NotImplementedError(
                f"RIFE architecture {arch_ver} is not implemented on the HIP path yet (supported: {SUPPORTED_ARCH})")
        if not torch.cuda.is_available():
            raise RuntimeError("RIFE VFI (HIP): no GPU visible; this node has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _lib.check(self.lib.vfi_init(self.device.index or 0), "vfi_init")
        check_state_dict(state_dict, arch_ver)
        self.arch_ver = arch_ver
        keys = list(rife_shapes(arch_ver).keys())
        tensors = [state_dict[k].detach().to("cpu", torch.float32).contiguous() for k in keys]
        ptrs = (C.c_void_p * len(keys))(*[t.data_ptr() for t in tensors])
        numels = (C.c_int64 * len(keys))(*[t.numel() for t in tensors])
        self.handle = self.lib.vfi_rife_create(ARCH_CODE[arch_ver], ptrs, numels, len(keys))
        if not self.handle:
            raise RuntimeError("vfi_rife_create failed: " + _lib.last_error())
        self.cfg = None

    @classmethod
    def clone_empty(cls, src, device):
        """The same network for another device of this process (current device must be ``device``): weight arena allocated
        but empty — multidev.RifeDeviceGroup fills it by one RCCL broadcast."""
        self = cls.__new__(cls)
        self.lib = src.lib
        self.device = torch.device(device)
        _lib.check(self.lib.vfi_init(self.device.index or 0), "vfi_init")
        self.arch_ver = src.arch_ver
        self.handle = self.lib.vfi_rife_clone_empty(src.handle)
        if not self.handle:
            raise RuntimeError("vfi_rife_clone_empty failed: " + _lib.last_error())
        self.cfg = None
        return self

    def weights(self):
        """(device pointer, float count) of the flat weight arena."""
        p, n = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.vfi_rife_weights(self.handle, C.byref(p), C.byref(n)), "vfi_rife_weights")
        return p.value, n.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vfi_rife_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, H, W, max_batch=1, n_slots=4, scale_factor=1.0):
        cfg = (H, W, max_batch, n_slots, float(scale_factor))
        if cfg != self.cfg:
            torch.cuda.synchronize(self.device)
            _lib.check(self.lib.vfi_rife_configure(self.handle, H, W, max_batch, n_slots, float(scale_factor)),
                       "vfi_rife_configure")
            self.cfg = cfg

    def load_frame(self, slot, frame_dev):
        """frame_dev: [H,W,C] device tensor (C >= 3), fp32 — or uint8, converted on the device (x / 255)."""
        assert frame_dev.is_cuda and frame_dev.dtype in (torch.float32, torch.uint8) and frame_dev.is_contiguous()
        assert tuple(frame_dev.shape[:2]) == self.cfg[:2], (frame_dev.shape, self.cfg)
        if frame_dev.dtype == torch.uint8:
            _lib.check(self.lib.vfi_rife_load_frame_u8(self.handle, slot, frame_dev.data_ptr(), frame_dev.shape[2],
                                                       _lib.stream_ptr()), "vfi_rife_load_frame_u8")
        else:
            _lib.check(self.lib.vfi_rife_load_frame(self.handle, slot, frame_dev.data_ptr(), frame_dev.shape[2],
                                                    _lib.stream_ptr()), "vfi_rife_load_frame")

    def load_frames(self, slots, frames_dev):
        """A batch of frames in one launch (arch 4.7; frame by frame on the others): frames_dev[i] -> slot slots[i]."""
        n = len(slots)
        if n == 0:
            return
        f0 = frames_dev[0]
        for f in frames_dev:
            assert f.is_cuda and f.dtype == f0.dtype and f.dtype in (torch.float32, torch.uint8) and f.is_contiguous()
            assert tuple(f.shape) == tuple(f0.shape) and tuple(f.shape[:2]) == self.cfg[:2], (f.shape, self.cfg)
        ptrs = (C.c_void_p * n)(*[f.data_ptr() for f in frames_dev])
        _lib.check(self.lib.vfi_rife_load_frames(self.handle, n, (C.c_int * n)(*slots), ptrs, f0.shape[2], int(f0.dtype == torch.uint8),
                                                 _lib.stream_ptr()), "vfi_rife_load_frames")

    def interpolate(self, slot0, slot1, timesteps, out_dev):
        """out_dev[b] = clamp(IFNet(frame[slot0[b]], frame[slot1[b]], t[b]), 0, 1);  out_dev [B,H,W,3]."""
        B = len(timesteps)
        assert out_dev.is_cuda and out_dev.dtype == torch.float32 and out_dev.is_contiguous()
        assert tuple(out_dev.shape) == (B, self.cfg[0], self.cfg[1], 3)
        s0 = (C.c_int * B)(*slot0)
        s1 = (C.c_int * B)(*slot1)
        ts = (C.c_float * B)(*[float(t) for t in timesteps])
        _lib.check(self.lib.vfi_rife_interpolate(self.handle, B, s0, s1, ts, out_dev.data_ptr(), _lib.stream_ptr()),
                   "vfi_rife_interpolate")

    def work_per_task(self):
        fl, by = C.c_double(), C.c_double()
        _lib.check(self.lib.vfi_rife_work(self.handle, C.byref(fl), C.byref(by)), "vfi_rife_work")
        return fl.value, by.value

    # -- test taps -----------------------------------------------------------------------
    def debug_keep(self, on=True):
        _lib.check(_lib.test_tap("vfi_rife_debug_keep")(self.handle, int(on)), "vfi_rife_debug_keep")

    def debug_read(self, what, stage, numel):
        buf = torch.empty(numel, dtype=torch.float32)
        n = _lib.test_tap("vfi_rife_debug_read")(self.handle, what, stage, buf.data_ptr(), numel)
        if n < 0:
            raise RuntimeError("vfi_rife_debug_read: " + _lib.last_error())
        return buf[:n]


class _FrameSlots:
    """Device frame cache bookkeeping: frame index -> slot, evicting frames no longer needed."""

    def __init__(self, n_slots):
        self.slot_of = {}
        self.free = list(range(n_slots))

    def assign(self, needed):
        """Make every frame of ``needed`` resident; returns the [(frame, slot)] that have to be (re)loaded, in order."""
        needed = list(dict.fromkeys(needed))
        missing = [f for f in needed if f not in self.slot_of]
        if len(missing) > len(self.free):  # recycle slots of frames this batch does not use
            for f in [f for f in self.slot_of if f not in needed]:
                self.free.append(self.slot_of.pop(f))
        if len(missing) > len(self.free):
            raise RuntimeError("frame cache too small for this batch")
        loads = []
        for f in missing:
            self.slot_of[f] = self.free.pop()
            loads.append((f, self.slot_of[f]))
        return loads

    def retire(self, dead):
        """Frames no later launch uses: their slots become free for frames packed ahead of their launch."""
        for f in dead:
            if f in self.slot_of:
                self.free.append(self.slot_of.pop(f))

    def take(self, f):
        """A free slot for frame ``f`` (packed ahead of the launch that needs it), or None."""
        if f in self.slot_of or not self.free:
            return None
        self.slot_of[f] = self.free.pop()
        return self.slot_of[f]


def launch_sizes(n, bs):
    """Tasks per launch for ``n`` tasks at ``bs`` per launch.  Clips of at least two launches start and end on a half-size one:
    the GPU starts after half as many uploads and the part of the download that nothing overlaps (the last launch's frames)
    is half as long — on a 33-frame fp32 1080p clip the device was busy from 7 to 69 ms of a 100 ms call."""
    half = bs // 2
    if half < 2 or n < 2 * bs:
        return [min(bs, n - pos) for pos in range(0, n, bs)]
    sizes, rest = [half], n - half
    while rest > bs + half:
        sizes.append(bs)
        rest -= bs
    return sizes + ([rest - half, half] if rest > bs else [rest])


def _batches(tasks, bs):
    pos = 0
    for size in launch_sizes(len(tasks), bs):
        bt = tasks[pos:pos + size]
        need = []
        for p, _ in bt:
            need += [p, p + 1]
        yield pos, bt, need
        pos += size


def _pick_loads(order, item, need_set, slots, ready, depth):
    """The next group of frames to pack, [(frame, slot)], taken from ``order[item:]`` in order: frames of ``need_set`` (this
    launch: always, the caller blocks on their upload), then frames of later launches while ``ready(i)`` says their upload has
    finished and fewer than PACK_AHEAD frames that this launch does not need are resident.  At most ``depth`` per group."""
    chunk = []
    while item + len(chunk) < len(order) and len(chunk) < depth:
        f = order[item + len(chunk)]
        must = f in need_set
        if not must and (not ready(item + len(chunk)) or sum(1 for g in slots.slot_of if g not in need_set) >= PACK_AHEAD):
            break
        slot = slots.take(f)      # (2 * bs + 2 slots are never taken by frames packed ahead: a needed frame always finds one)
        if slot is None:
            if must:
                raise RuntimeError("frame cache too small for this batch")
            break
        chunk.append((f, slot))
    return chunk


# (ckpt_name) -> RifeEngine; the reference caches by (ckpt, dtype, torch_compile), rife/__init__.py:31
_model_cache: typing.Dict[typing.Tuple, RifeEngine] = {}


_switch_lock, _switch_users, _switch_saved = threading.Lock(), 0, None


@contextlib.contextmanager
def _short_switch_interval(seconds=2e-4):
    """sys.setswitchinterval is process-wide: the first caller in saves the interpreter's value, the last one out restores it (two
    host threads running the node at once must not leave the process on the short interval for good)."""
    global _switch_users, _switch_saved
    import sys

    with _switch_lock:
        if _switch_users == 0:
            _switch_saved = sys.getswitchinterval()
            sys.setswitchinterval(seconds)
        _switch_users += 1
    try:
        yield
    finally:
        with _switch_lock:
            _switch_users -= 1
            if _switch_users == 0:
                sys.setswitchinterval(_switch_saved)


def run_tasks(engine, frames_cpu, tasks, batch_size, scale_factor=1.0, out_device=False, out=None, out_rows=None, on_staged=None):
    """Interpolate ``tasks`` = [(pair, t), ...] over host frames [N,H,W,C].

    Returns [len(tasks),H,W,3] (CPU tensor, or device tensor when ``out_device``).  With ``out``/``out_rows`` the
    new frame of task i is written straight into ``out[out_rows[i]]`` (the node's final output tensor) instead.

    Host pipeline (hostpipe.py): each input frame is uploaded once, through pinned staging and several frames ahead of
    the compute stream, and its pad/encode result is cached on the device — packed as soon as its upload has finished, up to
    PACK_AHEAD frames ahead of the launch that needs it; device outputs are double-buffered and batch k is moved to its final
    host rows by worker threads while batch k+1 computes.  ``on_staged()`` is called (from a worker thread) once the first
    launch's frames sit in pinned memory: the caller's other host-memory work (first touch of the output tensor, pass-through
    copies) starts then instead of competing with the copies the GPU is waiting for."""
    from .hostpipe import Downloader, Uploader, _T

    n, H, W, _ = frames_cpu.shape
    bs = max(1, min(int(batch_size), MAX_LIB_BATCH))
    # frame slots: what one launch can need (2 per task) + PACK_AHEAD frames whose upload has finished before their launch comes up
    # (a frame's pack depends on the frame alone: uploads and packs run ahead of the launches, which then never wait for PCIe)
    n_slots = 2 * bs + 2 + PACK_AHEAD
    engine.configure(H, W, bs, n_slots, scale_factor)
    dev = engine.device
    u8 = out is not None and out.dtype == torch.uint8      # 8-bit output rows: converted on the device, a quarter of the D2H bytes
    if out_device:
        res = torch.empty((len(tasks), H, W, 3), dtype=torch.float32, device=dev)
    elif out is None:
        res = torch.empty((len(tasks), H, W, 3), dtype=torch.float32)
        out, out_rows = res, list(range(len(tasks)))
    else:
        res = out
    main = torch.cuda.current_stream(dev)
    batches = list(_batches(tasks, bs))
    order, last_use = [], {}     # frames in order of first use = the upload sequence; the last launch that reads each
    for bi, (_, _, need) in enumerate(batches):
        for f in need:
            if f not in last_use:
                order.append(f)
            last_use[f] = bi
    if len(order) != len(set(order)):
        raise AssertionError("frame listed twice in the upload order")
    # A frame keeps its device slot from its pack to the last launch that reads it (uploaded once, never evicted).  For the task lists
    # this package builds — sorted by pair — at most 2 * bs + 1 frames are alive at a launch; a list that revisits old pairs can need
    # more than the slot pool holds, and must be told so before any work is queued rather than by a failed assertion in the loop.
    first_use = {}
    for bi, (_, _, need) in enumerate(batches):
        for f in need:
            first_use.setdefault(f, bi)
    delta = [0] * (len(batches) + 1)      # one sweep: +1 at a frame's first launch, -1 after its last (O(frames + launches))
    for f in order:
        delta[first_use[f]] += 1
        delta[last_use[f] + 1] -= 1
    alive = live = 0
    for d in delta[:-1]:
        live += d
        alive = max(alive, live)
    if alive > n_slots - PACK_AHEAD:
        raise ValueError(f"run_tasks: {alive} frames are alive at one launch but the frame cache holds {n_slots - PACK_AHEAD} (+ {PACK_AHEAD} packed ahead); "
                         f"pass the tasks sorted by pair (schedule.rife_task_list order) or a smaller batch_size")
    slots = _FrameSlots(n_slots)
    up = Uploader(frames_cpu, order, dev, main, depth=min(len(order), 2 * bs + 2 + PACK_AHEAD) or 1, on_staged=on_staged,
                  staged_after=min(len(order), len(dict.fromkeys(batches[0][2]))) if batches else 0)
    down = None if out_device else Downloader(dev, (H, W, 3), main, depth=2 * bs, dtype=torch.uint8 if u8 else torch.float32)
    bufs = [torch.empty((bs, H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)] if not out_device else None
    bufs8 = [torch.empty((bs, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)] if u8 else None
    buf_free = [None, None]
    import os, time
    tl = [] if HOST_TIMELINE else None
    if tl is not None:
        t_base = time.perf_counter()
        ev_base = torch.cuda.Event(enable_timing=True)
        ev_base.record(main)
    try:
        item, k = 0, 0       # item: next entry of `order` to pack
        for bi, (pos, bt, need) in enumerate(batches):
            if tl is not None:
                t_a = time.perf_counter() - t_base
            with _T("main.retire"):
                slots.retire([f for f, lu in last_use.items() if lu < bi])
            # Frames to pack now, in upload order: every frame this launch needs that is not resident (blocking on its upload), then —
            # without waiting — the frames of later launches whose upload has already finished, while slots are free.  ONE frame-pack
            # launch per group (vfi_rife_load_frames), in chunks of the staging ring's depth (a staging slot is recycled only after
            # its frame has been packed).
            need_set = set(need)
            while True:
                with _T("main.pick"):
                    chunk = _pick_loads(order, item, need_set, slots, up.ready, up.depth)
                if not chunk:
                    break
                with _T("main.get"):
                    srcs = [up.get(item + j) for j in range(len(chunk))]
                with _T("main.load_frame"):
                    engine.load_frames([slot for _, slot in chunk], srcs)
                with _T("main.release"):
                    for j in range(len(chunk)):
                        up.release(item + j)
                item += len(chunk)
                if all(f in slots.slot_of for f in need_set):
                    break
            missing = [f for f in need_set if f not in slots.slot_of]
            assert not missing, f"frames {missing} of launch {bi} are not resident"
            m = slots.slot_of
            if out_device:
                buf = res[pos:pos + len(bt)]
            else:
                buf = bufs[k][:len(bt)]
                if buf_free[k] is not None:
                    with _T("main.wait_buf"):
                        main.wait_event(buf_free[k])   # the copy-back of two batches ago has left this buffer
            if tl is not None:
                t_b = time.perf_counter() - t_base
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(main)
            with _T("main.interpolate"):
                engine.interpolate([m[p] for p, _ in bt], [m[p + 1] for p, _ in bt], [t for _, t in bt], buf)
            if tl is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(main)
                tl.append((t_a, t_b, time.perf_counter() - t_base, e0, e1))
            if not out_device:
                if u8:
                    _lib.check(engine.lib.vfi_f32_to_u8(buf.data_ptr(), bufs8[k].data_ptr(), buf.numel(), _lib.stream_ptr()), "vfi_f32_to_u8")
                    buf = bufs8[k][:len(bt)]
                done = torch.cuda.Event()
                done.record(main)
                with _T("main.push"):
                    buf_free[k] = down.push(done, buf, [out[out_rows[pos + i]] for i in range(len(bt))])
                if tl is not None:
                    tl[-1] = tl[-1] + (time.perf_counter() - t_base,)
                k ^= 1
    finally:
        if tl is not None:
            t_loop = time.perf_counter() - t_base
        up.close()
        if tl is not None:
            t_up = time.perf_counter() - t_base
        if down is not None:
            down.close()
        if tl is not None:
            t_down = time.perf_counter() - t_base
    if tl is not None:
        torch.cuda.synchronize()
        print(f"   tail: launch loop done {t_loop * 1e3:.1f} ms, uploader closed {t_up * 1e3:.1f}, downloader drained {t_down * 1e3:.1f}")
        print("   batch: host[begin, uploads ready, enqueued, copy-back enqueued] ms | device[start, end] ms (device clock zeroed at host 0)")
        for i, row in enumerate(tl):
            a, b, c, e0, e1 = row[:5]
            d = row[5] if len(row) > 5 else float("nan")
            print(f"   {i:3d}: host {a * 1e3:7.1f} {b * 1e3:7.1f} {c * 1e3:7.1f} {d * 1e3:7.1f} | device {ev_base.elapsed_time(e0):7.1f} {ev_base.elapsed_time(e1):7.1f}")
    return res


class RIFE_VFI:
    @staticmethod
    def _vfi40(engine, frames, multiplier, fast_mode, ensemble, scale_factor, batch_size, states):
        """arch 4.0: op-by-op engine (rife40.py).  The widget's batch_size is honoured exactly and tasks are not sharded over
        ranks: the scale list is doubled in place by a test over the whole batch (rife_arch.py:598-607) and stays doubled for
        the rest of the call, so batch composition and order are part of the result."""
        from .hostpipe import OutputWriter
        from .rife40 import run_tasks40

        frames = frames[..., :3]
        n = len(frames)
        _, tasks = rife_task_list(n, multiplier, states)
        plan = rife_output_plan(n, tasks)
        wr = OutputWriter(len(plan), frames.shape[1], frames.shape[2], engine.device)
        rows = [0] * len(tasks)
        for i, (kind, idx) in enumerate(plan):
            if kind == "new":
                rows[idx] = i
            else:
                wr.put_host(i, frames[idx])
        scale_list = [8 / scale_factor, 4 / scale_factor, 2 / scale_factor, 1 / scale_factor]   # rife/__init__.py:157-160
        keep = run_tasks40(engine, frames, tasks, batch_size, scale_list, fast_mode, ensemble, wr, rows)
        out = wr.finish()
        del keep
        print(f"Comfy-VFI done! {len(plan)} frames generated")
        return out

    @staticmethod
    def _device_group(cache_key, engine, arch_ver):
        """multidev.RifeDeviceGroup over the devices VFI_DEVICES / config.yaml select (None = just the engine's device)."""
        from . import multidev

        devices = multidev.selected_devices(engine.device)
        if len(devices) <= 1 or arch_ver == "4.0":
            return None
        key = cache_key + ("devices",) + tuple(devices)
        if key not in _model_cache:
            _model_cache[key] = multidev.RifeDeviceGroup.around(engine, devices)
        return _model_cache[key]

    @classmethod
    def INPUT_TYPES(s):
        return {
            "required": {
                "ckpt_name": (
                    sorted(list(CKPT_NAME_VER_DICT.keys()),
                           key=lambda ckpt_name: version.parse(CKPT_NAME_VER_DICT[ckpt_name])),
                    {"default": "rife49.pth"},
                ),
                "frames": ("IMAGE",),
                "clear_cache_after_n_frames": ("INT", {"default": 10, "min": 1, "max": 1000}),
                "multiplier": ("INT", {"default": 2, "min": 1}),
                "fast_mode": ("BOOLEAN", {"default": True}),
                "ensemble": ("BOOLEAN", {"default": True}),
                "scale_factor": ([0.25, 0.5, 1.0, 2.0, 4.0], {"default": 1.0}),
                "dtype": (DTYPE_OPTIONS, {"default": "float32"}),
                "torch_compile": ("BOOLEAN", {"default": False,
                                              "tooltip": "Ignored on the HIP path: kernels are compiled ahead of time."}),
                "batch_size": ("INT", {"default": 1, "min": 1, "max": 64,
                                       "tooltip": "Number of interpolation tasks per GPU call."}),
            },
            "optional": {"optional_interpolation_states": ("INTERPOLATION_STATES",)},
        }

    RETURN_TYPES = ("IMAGE",)
    FUNCTION = "vfi"
    CATEGORY = "ComfyUI-Frame-Interpolation/VFI"

    def vfi(
        self,
        ckpt_name: typing.AnyStr,
        frames: torch.Tensor,
        clear_cache_after_n_frames: int = 10,
        multiplier: typing.SupportsInt = 2,
        fast_mode: bool = False,
        ensemble: bool = False,
        scale_factor: float = 1.0,
        dtype: str = "float32",
        torch_compile: bool = False,
        batch_size: int = 1,
        optional_interpolation_states: InterpolationStateList = None,
        **kwargs,
    ):
        if dtype not in DTYPE_OPTIONS:
            raise KeyError(dtype)
        if dtype != "float32":
            if frames.dtype == torch.uint8:
                frames = frames.to(torch.float32) / 255.0
            # The reference casts model and inputs to the widget's dtype, rounds every output frame through it
            # (rife/__init__.py:120-134,195-198,210,227-230) and ALWAYS returns float32 (:237-238).  Here the hot path
            # computes in fp32: the clip is rounded to the requested dtype on the way in (so pass-through frames are
            # bit-identical to the reference's), new frames are the fp32 result rounded once through that dtype, and
            # the IMAGE tensor comes back as float32 like the reference's.
            torch_dtype = DTYPE_MAP[dtype]
            warnings.warn(f"RIFE VFI (HIP): dtype={dtype}: I/O in {dtype}, compute in float32.")
            (out,) = self.vfi(ckpt_name, frames.to(torch_dtype).to(torch.float32), clear_cache_after_n_frames, multiplier,
                              fast_mode, ensemble, scale_factor, "float32", torch_compile, batch_size,
                              optional_interpolation_states, **kwargs)
            return (out.to(torch_dtype).to(torch.float32),)
        arch_ver = CKPT_NAME_VER_DICT[ckpt_name]
        cache_key = (ckpt_name,)
        prewarm = None
        if cache_key not in _model_cache:
            if arch_ver != "4.0" and world()[1] == 1 and os.environ.get("VFI_DEVICES", "current") == "current" and torch.cuda.is_available() \
                    and frames.dim() == 4 and len(frames) >= 2:
                # the process's first call: pin the host staging rings beside the checkpoint load / weight pack (hostpipe.prewarm_rings_async)
                from .hostpipe import prewarm_rings_async

                _, tasks_ = rife_task_list(len(frames), multiplier, optional_interpolation_states)
                bs_ = effective_batch(batch_size, frames.shape[1], frames.shape[2], len(tasks_))
                u8_ = frames.dtype == torch.uint8
                prewarm = prewarm_rings_async(torch.device("cuda", torch.cuda.current_device()), (frames.shape[1], frames.shape[2], 3),
                                              torch.uint8 if u8_ else torch.float32, torch.uint8 if u8_ else torch.float32,
                                              min(len(frames), 2 * bs_ + 2 + PACK_AHEAD), 2 * bs_)
            model_path = load_file_from_github_release(MODEL_TYPE, ckpt_name)
            sd = torch.load(model_path, map_location="cpu", weights_only=False)
            if arch_ver == "4.0":
                from .rife40 import Rife40Engine
                _model_cache[cache_key] = Rife40Engine(sd)
            else:
                _model_cache[cache_key] = RifeEngine(sd, arch_ver)
            print(f"Comfy-VFI: Loaded and cached model {ckpt_name} (HIP, float32)")
        engine = _model_cache[cache_key]
        if arch_ver == "4.0":
            if frames.dtype == torch.uint8:
                frames = frames.to(torch.float32) / 255.0      # the op-by-op 4.0 engine takes float32 clips only
            return (self._vfi40(engine, frames, multiplier, fast_mode, ensemble, scale_factor, batch_size,
                                optional_interpolation_states),)

        frames = frames[..., :3]  # preprocess_frames: drop alpha; layout stays NHWC on this path
        n = len(frames)
        _, tasks = rife_task_list(n, multiplier, optional_interpolation_states)
        plan = rife_output_plan(n, tasks)
        # Extension beyond the reference: a uint8 clip (what video load / save nodes hold) stays 8-bit
        # on the host and over PCIe — x / 255 on the device on the way in, round(clamp(y) * 255) on the way out, uint8 IMAGE
        # returned; the float32 contract of the reference is untouched for float32 clips.
        u8 = frames.dtype == torch.uint8
        if u8 and world()[1] > 1:
            raise NotImplementedError("RIFE VFI (HIP): uint8 clips are not supported on the torch.distributed multi-process path")
        if not u8 and frames.dtype != torch.float32:
            frames = frames.to(torch.float32)
        out = torch.empty((len(plan),) + tuple(frames.shape[1:]), dtype=torch.uint8 if u8 else torch.float32)
        src_rows = [i for i, (kind, _) in enumerate(plan) if kind == "src"]
        src_idx = [idx for kind, idx in plan if kind == "src"]
        new_rows = [0] * len(tasks)
        for i, (kind, idx) in enumerate(plan):
            if kind == "new":
                new_rows[idx] = i
        # pass-through frames (bit-exact, alpha already dropped): copied by background threads while the GPU works
        from .hostpipe import copy_rows_async, prefault_async
        # first touch of the fresh output tensor by a few background threads (see hostpipe.py), in address order and
        # ahead of the copies that fill it
        # (starting them only after the first launch was tried: the GPU starts 8 ms earlier and the call ends 15 ms later —
        # the downloads then wait for pages)
        passthrough, started, start_lock = [], [False], threading.Lock()

        def start_host_side():      # first touch of the output + pass-through copies: ~200 ms of host-thread time on a 33-frame 1080p clip
            # called from an uploader worker (on_staged) AND from this thread after the launch loop: test-and-set and the submissions under
            # one lock, so the work is submitted exactly once and whoever comes second returns only after `passthrough` is complete
            with start_lock:
                if not started[0]:
                    started[0] = True
                    passthrough.extend(prefault_async(out))
                    passthrough.extend(copy_rows_async(out, src_rows, frames, src_idx))

        batch_size = effective_batch(batch_size, frames.shape[1], frames.shape[2], len(tasks))
        rank, ws = world()
        group = self._device_group(cache_key, engine, arch_ver) if ws == 1 else None
        single = group is None and ws == 1 and len(tasks) > 0
        if not single:
            start_host_side()
        if group is not None:
            # one process, several GPUs (multidev.py): every device interpolates its block of the task list and copies its own
            # shard into `out` over its own PCIe link
            group.run(frames, tasks, batch_size, scale_factor, out, new_rows)
        elif ws > 1:
            lo, hi = shard_tasks(tasks, rank, ws)
            counts = [shard_tasks(tasks, r, ws)[1] - shard_tasks(tasks, r, ws)[0] for r in range(ws)]
            local = run_tasks(engine, frames, tasks[lo:hi], batch_size, scale_factor, out_device=True)
            new_frames = all_gather_frames(local, counts)
            from .hostpipe import Downloader
            main = torch.cuda.current_stream(engine.device)
            down = Downloader(engine.device, tuple(frames.shape[1:3]) + (3,), main, depth=16)
            ready = torch.cuda.Event()
            ready.record(main)
            for c in range(0, len(tasks), 8):
                down.push(ready, new_frames[c:c + 8], [out[new_rows[i]] for i in range(c, min(c + 8, len(tasks)))])
            down.close()
        else:
            # (started once the first launch's frames are in pinned memory: beside them the staging copies the GPU waits for took
            # 13 ms instead of 3 on the 256-thread host; started only after the first LAUNCH the downloads then wait for pages)
            # ~25 worker threads move frames while this thread feeds the GPU: at CPython's default 5 ms switch interval a thread that
            # wants the GIL can wait that long for it — the launch loop lost 10 ms between two launches that way
            # (profiles/r04_e2e_timeline.txt)
            if prewarm is not None:
                prewarm.join()
            with _short_switch_interval():
                run_tasks(engine, frames, tasks, batch_size, scale_factor, out=out, out_rows=new_rows, on_staged=start_host_side)
            start_host_side()
        from .hostpipe import _T
        with _T("main.wait_pass"):
            for f in passthrough:
                f.result()
        print(f"Comfy-VFI done! {len(plan)} frames generated")
        return (out,)
