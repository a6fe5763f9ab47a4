"""One process, several GPUs: the multi-GPU form of the node that IS a drop-in.

The reference's contract is one ``vfi()`` call in one ComfyUI process (__init__.py:24-48, vfi_models/rife/__init__.py:77-91).
So here one process drives every selected device: one host thread and one stream per device, the ``(pair, t)`` task list
block-partitioned over the devices (tasks are independent, rife/__init__.py:164-207), every device uploading only the frames
its block touches and copying ITS OWN shard of new frames straight into the shared host output tensor over its own PCIe
link — no device holds or downloads another device's frames.  RCCL over xGMI (csrc/comm.hip, ncclCommInitAll) carries the
weights, once, as one flat buffer from the device that packed them; the all-gather of new frames exists for device-side
consumers (bench.py measures it) and is not needed for the node's host output.

Opt-in: ``VFI_DEVICES=all`` or ``VFI_DEVICES=0,1,2,3`` (or ``devices:`` in config.yaml); default = the current device only
(ComfyUI owns device placement; grabbing every GPU of the box unasked would not be a drop-in).
The one-process-per-GPU ``torch.distributed`` path (dist.py) remains as the fallback for launchers that start N processes.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from .schedule import shard_tasks


def selected_devices(default_device=None):
    """HIP device ids this process should drive, first = the primary (it packs the weights and is the broadcast root)."""
    spec = os.environ.get("VFI_DEVICES")
    if spec is None:
        try:
            from .ckpt import load_config
            spec = str(load_config().get("devices", "current"))
        except Exception:
            spec = "current"
    spec = spec.strip().lower()
    n = torch.cuda.device_count()
    cur = torch.cuda.current_device() if default_device is None else torch.device(default_device).index or 0
    if spec in ("", "current", "none", "1gpu"):
        return [cur]
    if spec == "all":
        ids = list(range(n))
    else:
        ids = [int(x) for x in spec.replace(" ", "").split(",") if x != ""]
    bad = [d for d in ids if d < 0 or d >= n]
    if bad or len(set(ids)) != len(ids):
        raise ValueError(f"VFI_DEVICES={spec!r}: devices {bad or ids} not valid ({n} visible, ids must be distinct)")
    # The caller's device (where the cached engine and its weight arena live) is the primary: the RCCL broadcast root must be a
    # member of the clique.  An explicit list that leaves it out is refused — silently adding a GPU the user excluded (it may be
    # busy or reserved) is worse than an error that says what to change.
    if cur not in ids:
        raise ValueError(f"VFI_DEVICES={spec!r} leaves out device {cur}, where this node's model lives (ComfyUI's current device): "
                         f"add {cur} to the list, or make one of {ids} the current device before the node runs")
    ids.remove(cur)
    ids.insert(0, cur)
    return ids


def shard_bounds(n_tasks_or_tasks, n_devices):
    """[(lo, hi)] per device: the contiguous block partition of schedule.shard_tasks (blocks differ by at most one task,
    concatenating them in device order restores the task order; a device may get an empty block)."""
    tasks = n_tasks_or_tasks if hasattr(n_tasks_or_tasks, "__len__") else range(n_tasks_or_tasks)
    return [shard_tasks(tasks, r, n_devices) for r in range(n_devices)]


def run_sharded(devices, n_tasks, fn, set_device=True):
    """Run ``fn(rank, device, lo, hi)`` for every device's block on its own host thread (the calling thread takes rank 0's
    block itself) and wait for all; the first exception of any thread is re-raised after every thread has finished.
    Devices with an empty block are not started."""
    bounds = shard_bounds(n_tasks, len(devices))
    errors = [None] * len(devices)

    def body(r):
        lo, hi = bounds[r]
        try:
            if set_device:
                torch.cuda.set_device(devices[r])        # per-thread current device (HIP and torch)
            fn(r, devices[r], lo, hi)
        except BaseException as e:  # noqa: BLE001  (re-raised below, on the caller's thread)
            errors[r] = e

    threads = []
    for r in range(1, len(devices)):
        if bounds[r][1] > bounds[r][0]:
            t = threading.Thread(target=body, args=(r,), name=f"vfi-dev{devices[r]}", daemon=True)
            t.start()
            threads.append(t)
    prev = torch.cuda.current_device() if set_device else None
    try:
        if bounds[0][1] > bounds[0][0]:
            body(0)
    finally:
        for t in threads:
            t.join()
        if set_device:
            torch.cuda.set_device(prev)
    for e in errors:
        if e is not None:
            raise e
    return bounds


class Comm:
    """RCCL clique of this process's devices (ncclCommInitAll) + one communication stream per device."""

    def __init__(self, devices):
        self.lib = _lib.load()
        self.devices = list(devices)
        arr = (C.c_int * len(devices))(*devices)
        self.handle = self.lib.vfi_comm_create(len(devices), arr)
        if not self.handle:
            raise RuntimeError("vfi_comm_create failed: " + _lib.last_error())
        self.streams = [torch.cuda.Stream(device=d) for d in devices]

    def _ptrs(self, vals):
        return (C.c_void_p * len(vals))(*vals)

    def broadcast(self, dev_ptrs, count, root=0):
        _lib.check(self.lib.vfi_comm_broadcast(self.handle, self._ptrs(dev_ptrs), count, root,
                                               self._ptrs([s.cuda_stream for s in self.streams])), "vfi_comm_broadcast")

    def all_gather_v(self, dev_ptrs, counts):
        _lib.check(self.lib.vfi_comm_all_gather_v(self.handle, self._ptrs(dev_ptrs), (C.c_int64 * len(counts))(*counts),
                                                  self._ptrs([s.cuda_stream for s in self.streams])), "vfi_comm_all_gather_v")

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vfi_comm_destroy(self.handle)
            self.handle = None


class RifeDeviceGroup:
    """The same RIFE network resident on every device of ``devices``: packed once (primary device), cloned empty on the
    others, weights broadcast as one flat buffer over RCCL."""

    def __init__(self, state_dict, arch_ver, devices, _primary=None):
        from .rife import RifeEngine

        self.devices = list(devices)
        self.lib = _lib.load()
        prev = torch.cuda.current_device()
        self.comm = None
        self.engines = []
        self.owns_primary = _primary is None
        try:
            torch.cuda.set_device(self.devices[0])
            self.engines.append(_primary if _primary is not None else
                                RifeEngine(state_dict, arch_ver, device=torch.device("cuda", self.devices[0])))
            if len(self.devices) > 1:
                self.comm = Comm(self.devices)
                for d in self.devices[1:]:
                    torch.cuda.set_device(d)
                    self.engines.append(RifeEngine.clone_empty(self.engines[0], torch.device("cuda", d)))
                ptrs, count = [], None
                for e in self.engines:
                    p, n = e.weights()
                    ptrs.append(p)
                    count = n if count is None else count
                    assert n == count
                torch.cuda.synchronize(self.devices[0])          # the primary's upload is complete
                self.comm.broadcast(ptrs, count, root=0)
                self.comm.synchronize()
        finally:
            torch.cuda.set_device(prev)

    @classmethod
    def around(cls, primary_engine, devices):
        """Group whose first member is an existing engine (the node's cached one); devices[0] must be its device."""
        if not devices or devices[0] != (primary_engine.device.index or 0):
            raise ValueError(f"RifeDeviceGroup.around: devices {list(devices)} must start with the primary engine's device "
                             f"{primary_engine.device.index or 0} (its weight arena is the broadcast root)")
        return cls(None, primary_engine.arch_ver, devices, _primary=primary_engine)

    @property
    def device(self):
        return self.engines[0].device

    def run(self, frames_cpu, tasks, batch_size, scale_factor, out, out_rows):
        """Interpolate ``tasks`` over the devices; new frame of task i lands in ``out[out_rows[i]]`` (host tensor), copied by
        the device that computed it."""
        from .rife import run_tasks

        def one(r, dev, lo, hi):
            run_tasks(self.engines[r], frames_cpu, tasks[lo:hi], batch_size, scale_factor, out=out, out_rows=out_rows[lo:hi])
            torch.cuda.synchronize(dev)

        return run_sharded(self.devices, len(tasks), one)

    def close(self):
        for e in self.engines[0 if self.owns_primary else 1:]:
            e.close()
        self.engines = []
        if self.comm is not None:
            self.comm.close()
            self.comm = None
