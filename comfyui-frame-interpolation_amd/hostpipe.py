"""Host <-> device frame pipeline for the node classes.

The node contract is host tensors in, a host tensor out (vfi_models/rife/__init__.py:195-207,225-239 moves every
batch with blocking ``.to(device)`` / ``.cpu()`` calls).  At ~2 ms of device time per 1080p frame those blocking pageable
copies — and the first-touch page faults of the freshly allocated output tensor — are what bounds the node, so:

  * uploads:   worker threads copy frame f into a pinned staging slot (this also performs the alpha drop for RGBA
               clips), then issue the H2D DMA on an upload stream, several frames ahead of the compute stream;
  * downloads: the compute stream's results are DMA'd into pinned slots on a download stream; worker threads move
               them into their final rows of the output tensor (parallel first-touch) while the next batch computes;
  * pass-through frames are copied by a third pool concurrently with everything else.

torch is used for what it is here for: pinned allocations, streams, events.  Pinned rings are cached per
(device, frame shape) for the life of the process — pinning costs about as much as the copies it saves.
"""
import ctypes
import os
import threading
from concurrent.futures import Future, ThreadPoolExecutor

import torch

_ring_cache = {}
_pools = {}
# worker threads per pool (upload staging, download drain, pass-through copies, first-touch).  The staging copies are plain
# memmoves at ~8-10 GB/s per thread; on the 256-thread host of the MI355X box 6,8,4,8 measured 308-319 interpolated frames/s end
# to end against 270-285 with 2,4,2,6 (33-frame 1080p clip) — small hosts keep the small pools.
WORKERS = tuple(int(x) for x in os.environ.get("VFI_HOST_WORKERS", "6,8,4,8" if (os.cpu_count() or 1) >= 32 else "2,4,2,6").split(","))


# optional wall-clock accounting per phase (VFI_HOST_PROFILE=1): seconds summed over worker / main-thread calls
PROFILE = os.environ.get("VFI_HOST_PROFILE", "0") == "1"
stats = {}
_stats_lock = threading.Lock()


class _T:
    def __init__(self, key):
        self.key = key

    def __enter__(self):
        if PROFILE:
            import time
            self.t0 = time.perf_counter()

    def __exit__(self, *a):
        if PROFILE:
            import time
            dt = time.perf_counter() - self.t0
            with _stats_lock:
                c = stats.setdefault(self.key, [0, 0.0])
                c[0] += 1
                c[1] += dt


SPLIT_PARTS = 4   # row bands per frame of the first launch's uploads (1 = whole frames)
POLL = False      # diagnostics (tools/node_e2e.py sets it): busy-poll the events instead of blocking in event.synchronize()


def _wait(ev):
    """Wait for a device event from a worker thread."""
    if POLL:
        import time
        while not ev.query():
            time.sleep(0.0002)
    else:
        ev.synchronize()


def host_copy(dst, src):
    """dst.copy_(src) for host tensors without torch's intra-op thread team: from a worker thread every torch copy
    would spin up its own OpenMP team (hundreds of threads on a big host); a plain memmove releases the GIL and lets
    the pools provide the parallelism.  Falls back to copy_ for strided sources (RGBA clips: the alpha drop)."""
    if dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel():
        ctypes.memmove(dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size())
    else:
        dst.copy_(src)


def _events(n, stream):
    """n events whose native handles already exist (torch creates them at the first record): made BEFORE the output tensor's first
    touch starts — while 8 threads populate pages, anything in the process that maps memory (event creation does) queues behind them:
    the first copy-back of a call took 4.5 ms instead of 0.5."""
    evs = [torch.cuda.Event() for _ in range(n)]
    for e in evs:
        e.record(stream)
    return evs


def _ring_events(r, key, n, stream):
    """n events kept with the staging set `r` (exclusively the caller's between acquire and release) and reused by every call: a 33-frame
    clip needs ~80 of them, and creating + destroying that many per call cost ~1 ms at the head and ~2 ms after the last copy-back."""
    have = r.setdefault(key, [])
    if len(have) < n:
        have.extend(_events(n - len(have), stream))
    return have[:n]


def _memcpy_async(dst, src, kind, stream):
    """dst.copy_(src, non_blocking=True) between a pinned host tensor and a device tensor (same dtype, both contiguous) as ONE
    hipMemcpyAsync through the library (vfi_memcpy_async): torch's copy_ holds the interpreter lock through its dispatch and
    queries the pointers' attributes — 0.6-2.7 ms per call beside 25 worker threads, on the launch loop's critical path."""
    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()
    from . import _lib

    _lib.check(_lib.load().vfi_memcpy_async(dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size(), kind, ctypes.c_void_p(stream.cuda_stream)),
               "vfi_memcpy_async")


_stream_cache = {}


def _stream(device, role):
    """The pipeline's copy streams, made once per (device, role) and reused by every call: torch hands out side streams from a pool
    round-robin and the HIP runtime maps streams onto a handful of hardware queues, so a stream made per call shared a queue with
    the compute stream in every other call — its copies then queued behind the kernels (node calls alternated 53 / 63 ms)."""
    key = (str(device), role)
    st = _stream_cache.get(key)
    if st is None:
        # r6: a library-made stream probed (vfi_stream_spin) to sit on another hardware queue than the compute stream and than the
        # pipeline's other copy stream — "from a pool, round robin" left that to chance
        from . import _lib

        busy = [torch.cuda.current_stream(device)] + [v for (d, _), v in _stream_cache.items() if d == str(device)]
        try:
            own = _lib.own_streams_apart(device, 1, avoid=busy)[0]      # (wishes are dropped from the end when the queues run out: the compute stream last)
            _stream_owned.append(own)
            st = own.stream
        except Exception:      # noqa: BLE001 — a plain pooled stream does the same work
            st = torch.cuda.Stream(device)
        _stream_cache[key] = st
    return st


_stream_owned = []


def _pool(name, n):
    if name not in _pools:
        _pools[name] = ThreadPoolExecutor(max_workers=n, thread_name_prefix="vfi-" + name)
    return _pools[name]


_ring_lock = threading.Lock()


def _rings_acquire(device, shape, dtype=torch.float32, role="up"):
    """A set of pinned / device staging rings for this (device, frame shape, role), exclusively the caller's until
    ``_rings_release``; sets are kept for the life of the process (pinning costs about as much as the copies it saves) and each
    ring only ever grows.  Concurrent pipelines on one device (two host threads) get different sets.  ``role`` ("up" / "down")
    keeps the uploader's and the downloader's sets apart: sharing one free list made each pick up the other's set on the next
    call and grow the rings it lacked — twice the pinned and device staging memory in steady state."""
    key = (str(device), tuple(shape), dtype, role)
    with _ring_lock:
        free = _ring_cache.setdefault(key, [])
        r = free.pop() if free else {"key": key, "up_host": [], "up_dev": [], "down_host": []}
    busy = r.pop("busy", None)
    if busy is not None:
        busy.synchronize()        # the previous owner's kernels may still be reading the device slots
    return r


def _rings_release(r):
    if r is not None:
        with _ring_lock:
            _ring_cache.setdefault(r["key"], []).append(r)


def _rings_grow(r, device, shape, n_up, n_down):
    dtype = r["key"][2]
    try:
        while len(r["up_host"]) < n_up:
            r["up_host"].append(torch.empty(shape, dtype=dtype, pin_memory=True))
            r["up_dev"].append(torch.empty(shape, dtype=dtype, device=device))
        while len(r["down_host"]) < n_down:
            r["down_host"].append(torch.empty(shape, dtype=dtype, pin_memory=True))
    except BaseException:
        r["up_host"] = r["up_host"][:len(r["up_dev"])]      # keep the pairs aligned, give the set back: an allocation failure must
        _rings_release(r)                                    # not strand it outside the cache
        raise
    return r


def prewarm_rings_async(device, shape, up_dtype, down_dtype, n_up, n_down):
    """First call of a process: pin the staging rings on a side thread WHILE the caller loads and packs the checkpoint (pinning ~1 GB
    is as long as building the engine: the two used to run back to back).  Returns the thread; the caller joins it before it builds
    its Uploader / Downloader, which then find the grown sets in the cache."""
    def work():
        try:
            torch.cuda.set_device(device)
            up = _rings_grow(_rings_acquire(device, shape, up_dtype, "up"), device, shape, n_up, 0)
            _rings_release(up)
            down = _rings_grow(_rings_acquire(device, shape, down_dtype, "down"), device, shape, 0, n_down)
            _rings_release(down)
        except Exception:      # best effort: the pipelines allocate what is missing themselves
            pass

    t = threading.Thread(target=work, name="vfi-prewarm", daemon=True)
    t.start()
    return t


class Uploader:
    """Stages ``frames[order[i]]`` (host, [H,W,C>=3]) to the device ahead of use.  ``get(i)`` returns a device tensor
    [H,W,3] valid on ``main`` until ``release(i)``; items must be consumed in order."""

    def __init__(self, frames, order, device, main, depth=6, workers=WORKERS[0], on_staged=None, staged_after=0):
        self.frames, self.order, self.device, self.main = frames, list(order), device, main
        # on_staged(): called once, from a worker, when the first `staged_after` items have been copied into pinned memory
        self._on_staged, self._staged_left, self._staged_lock = on_staged, int(staged_after), threading.Lock()
        self._staged_n = int(staged_after)
        if on_staged is not None and self._staged_left <= 0:
            on_staged()
            self._on_staged = None
        H, W = frames.shape[1:3]
        self.depth = depth
        self.rings = _rings_grow(_rings_acquire(device, (H, W, 3), frames.dtype, "up"), device, (H, W, 3), depth, 0)   # fp32 or uint8 clips
        self.host, self.dev = self.rings["up_host"][:depth], self.rings["up_dev"][:depth]
        self.stream = _stream(device, "up")
        self.freed = [threading.Event() for _ in self.order]      # slot of item i may be overwritten
        self.consumed = [None] * len(self.order)                  # cuda event: main stream finished reading item i
        # (an item's events are dead once item i + depth has been staged: a ring of 4 * depth serves clips of any length)
        self._ne = max(1, min(len(self.order), 4 * depth))
        self._ev_up = _ring_events(self.rings, "ev_up", self._ne, self.stream)      # H2D of item i done
        self._ev_cons = _ring_events(self.rings, "ev_cons", self._ne, main)         # main stream has consumed item i
        pool = _pool(f"up{device}", workers)
        # The frames of the first launch (`staged_after` of them) are what the GPU waits for at the head of a call: each is copied into
        # its pinned slot in SPLIT_PARTS row bands by as many workers, so that frame 0's H2D starts after a quarter of a frame copy
        # and the others follow back to back (whole frames per worker: all of them finish together, then queue on the link).
        self._split_n = min(int(staged_after), len(self.order), depth) if SPLIT_PARTS > 1 else 0
        self._parts_left = [SPLIT_PARTS] * self._split_n
        self.futs = []
        for i in range(len(self.order)):
            if i < self._split_n:
                fut = Future()
                fut.set_running_or_notify_cancel()
                self.futs.append(fut)
                for part in range(SPLIT_PARTS):
                    pool.submit(self._stage_part, i, part, fut)
            else:
                self.futs.append(pool.submit(self._stage, i))

    def _staged_one(self, i):
        """Item i sits in pinned memory: fire on_staged once the first `staged_after` items do."""
        if self._on_staged is not None and i < self._staged_n:
            fire = False
            with self._staged_lock:
                self._staged_left -= 1
                if self._staged_left == 0 and self._on_staged is not None:
                    fire = True
            if fire:
                cb, self._on_staged = self._on_staged, None
                cb()

    def _enqueue_h2d(self, i):
        s = i % self.depth
        ev = self._ev_up[i % self._ne]
        with _T("up.enqueue_h2d"):
            _memcpy_async(self.dev[s], self.host[s], 1, self.stream)
            ev.record(self.stream)
        return ev

    def _stage_part(self, i, part, fut):
        """Row band `part` of a first-launch frame (i < depth: its slot is free from the start); the band that finishes last issues the H2D."""
        try:
            s = i % self.depth
            src = self.frames[self.order[i]]
            rows = src.shape[0]
            r0, r1 = rows * part // SPLIT_PARTS, rows * (part + 1) // SPLIT_PARTS
            with _T("up.memcpy_to_pinned"):
                host_copy(self.host[s][r0:r1], src[r0:r1][..., :3])
            with self._staged_lock:
                self._parts_left[i] -= 1
                last = self._parts_left[i] == 0
            if last:
                self._staged_one(i)
                fut.set_result(self._enqueue_h2d(i))
        except BaseException as e:      # surfaces in get(i) / close()
            if not fut.done():
                fut.set_exception(e)

    def _stage(self, i):
        s = i % self.depth
        if i >= self.depth:
            with _T("up.wait_slot"):
                self.freed[i - self.depth].wait()
                _wait(self.consumed[i - self.depth])
        with _T("up.memcpy_to_pinned"):
            host_copy(self.host[s], self.frames[self.order[i]][..., :3])   # pageable -> pinned (drops alpha, makes contiguous)
        self._staged_one(i)
        return self._enqueue_h2d(i)

    def ready(self, i):
        """Has item i been staged and its H2D copy been enqueued?  (Non-blocking: frames packed ahead of their launch.)"""
        return self.futs[i].done()

    def get(self, i, stream=None):
        """Device tensor of item i, valid on ``stream`` (default: the main stream).  A second reader on another stream (pair lanes:
        frame p+1 of pair p is frame p of pair p+1) calls get() again with its own stream."""
        with _T("main.wait_upload"):
            ev = self.futs[i].result()
        (stream if stream is not None else self.main).wait_event(ev)
        return self.dev[i % self.depth]

    def release(self, i):
        ev = self._ev_cons[i % self._ne]
        ev.record(self.main)
        self.consumed[i] = ev
        self.freed[i].set()

    def close(self):
        for i, f in enumerate(self.futs):   # unblock and surface worker errors
            if self.consumed[i] is None:
                self.consumed[i] = torch.cuda.Event()
                self.consumed[i].record(self.main)
            self.freed[i].set()
        try:
            for f in self.futs:
                f.result()
        finally:
            if self.rings is not None:
                busy = torch.cuda.Event()
                busy.record(self.main)
                self.rings["busy"] = busy
            _rings_release(self.rings)
            self.rings = None


class Downloader:
    """Moves device frames [H,W,3] into rows of a host tensor through pinned slots, off the critical path."""

    def __init__(self, device, shape, main, depth=16, workers=WORKERS[1], dtype=torch.float32):
        self.device, self.main, self.depth = device, main, depth
        self.rings = _rings_grow(_rings_acquire(device, shape, dtype, "down"), device, shape, 0, depth)
        self.host = self.rings["down_host"][:depth]
        self.stream = _stream(device, "down")
        self.pool = _pool(f"down{device}", workers)
        self.workers = workers
        self._ev = _ring_events(self.rings, "ev_down", depth, self.stream)      # one per staging slot (a slot's previous copy has been drained before reuse)
        self.slot_fut = [None] * depth
        self.n = 0
        self.futs = []

    def _finish(self, ev, s, dst, part=0, nparts=1):
        with _T("down.wait_d2h"):
            _wait(ev)
        with _T("down.memcpy_to_out"):
            if nparts == 1:
                host_copy(dst, self.host[s])
            else:       # rows [r0, r1) of the frame: the last launch's few frames are moved by all workers, not one worker per frame
                rows = dst.shape[0]
                r0, r1 = rows * part // nparts, rows * (part + 1) // nparts
                host_copy(dst[r0:r1], self.host[s][r0:r1])

    def push(self, ready_event, dev_frames, dst_rows):
        """After ``ready_event`` (recorded on the compute stream), copy dev_frames[i] -> dst_rows[i].
        Returns a cuda event that fires when the device buffer has been read completely."""
        evs = []
        nparts = max(1, min(4, self.workers // max(1, len(dst_rows)))) if len(dst_rows) else 1
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready_event)
            for i, dst in enumerate(dst_rows):
                s = self.n % self.depth
                if self.slot_fut[s] is not None:
                    with _T("main.wait_down_slot"):
                        for f in self.slot_fut[s]:
                            f.result()                 # slot still being drained by a worker
                _memcpy_async(self.host[s], dev_frames[i], 2, self.stream)
                ev = self._ev[s]
                ev.record(self.stream)
                fs = [self.pool.submit(self._finish, ev, s, dst, part, nparts) for part in range(nparts)]
                self.slot_fut[s] = fs
                self.futs.extend(fs)
                self.n += 1
                evs.append(ev)
        return evs[-1] if evs else ready_event

    def close(self):
        try:
            with _T("main.wait_down_close"):
                for f in self.futs:
                    f.result()
        finally:
            self.futs = []
            _rings_release(self.rings)
            self.rings = None


def copy_rows_async(out, rows, frames, idx, workers=WORKERS[2]):
    """out[rows[i]] = frames[idx[i]][..., :3] on a background pool (pass-through frames); returns futures."""
    pool = _pool("pass", workers)

    def one(r, j):
        with _T("pass.memcpy"):
            host_copy(out[r], frames[j][..., :3])

    return [pool.submit(one, r, j) for r, j in zip(rows, idx)]


# ---- first-touch of the output tensor --------------------------------------------------------------------------
# A fresh torch.empty() of N_out x 24.9 MB is untouched anonymous memory: every 4 KiB written for the first time is a
# page fault (measured 2.9 GB/s per thread on the GPU box = 8.5 ms per 1080p frame, 4x the device time).  The output
# is therefore populated up front by a few threads with madvise(MADV_HUGEPAGE) + madvise(MADV_POPULATE_WRITE)
# (Linux >= 5.14; skipped on older kernels), overlapping the uploads and the first batches.
_MADV_HUGEPAGE, _MADV_POPULATE_WRITE = 14, 23
_libc = None


def _madvise(addr, nbytes, advice):
    global _libc
    if _libc is None:
        _libc = ctypes.CDLL(None, use_errno=True)
        _libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    a = (addr + 4095) & ~4095
    n = (addr + nbytes - a) & ~4095
    return (a, n, _libc.madvise(a, n, advice)) if n > 0 else (a, 0, 0)


_populate_ok = True


def _populate(addr, nbytes):
    """Runs concurrently with the writers of the same tensor, so it must never modify contents: MADV_POPULATE_WRITE
    faults pages in without touching data.  Where the kernel lacks it (< 5.14: EINVAL) the prefault is simply skipped
    — the writers then take the first-touch faults themselves (slower, never wrong)."""
    global _populate_ok
    if not _populate_ok:
        return
    with _T("prefault"):
        a, n, rc = _madvise(addr, nbytes, _MADV_POPULATE_WRITE)
        if n > 0 and rc != 0:
            _populate_ok = False


def prefault_async(t, chunk=2 << 20, workers=None):
    """Populate the pages of host tensor ``t`` (contents undefined afterwards, like torch.empty) in the background,
    in address order.  MADV_HUGEPAGE changes VMA flags (mmap write lock), so it is issued once, up front; the
    populate calls only take the read side and are kept SHORT (2 MiB, ~0.25 ms): anything in the process that needs the
    write side — the HIP runtime mapping memory for the launching thread — queues behind the populate calls in flight, and
    with 16 MiB calls the launch loop stood still for ~14 ms right after the prefault started (profiles/r04_e2e_timeline.txt)."""
    workers = workers if workers is not None else (WORKERS[3] if len(WORKERS) > 3 else 6)
    if workers <= 0 or not t.is_contiguous() or t.device.type != "cpu":
        return []
    pool = _pool("fault", workers)
    base, n = t.data_ptr(), t.numel() * t.element_size()
    if os.environ.get("VFI_HOST_THP", "1") == "1":
        _madvise(base, n, _MADV_HUGEPAGE)   # best effort (THP may be disabled)
    n_chunks = (n + chunk - 1) // chunk

    def sweep(k):      # chunks k, k + workers, ...: together the tasks walk the tensor in address order; ONE future per worker
        for c in range(k, n_chunks, workers):     # (772 submissions of 2 MiB each cost the submitting thread ~20 ms of GIL time)
            off = c * chunk
            _populate(base + off, min(chunk, n - off))

    return [pool.submit(sweep, k) for k in range(min(workers, n_chunks))]


class OutputWriter:
    """The node's output tensor [rows,H,W,3] (host, fp32) filled in the background: pass-through frames from host
    tensors, new frames from device tensors.  ``finish()`` waits for everything and returns the tensor."""

    def __init__(self, rows, H, W, device, depth=8):
        self.out = torch.empty((rows, H, W, 3), dtype=torch.float32)
        self.futs = prefault_async(self.out)
        self.device = device
        self.main = torch.cuda.current_stream(device)
        self.down = Downloader(device, (H, W, 3), self.main, depth=depth)

    def put_host(self, row, frame):
        self.futs += copy_rows_async(self.out, [row], frame[None], [0])

    def put_dev(self, row, dev_frame, stream=None):
        """dev_frame [H,W,3] (written on ``stream``, default the main stream) must stay untouched until the returned event has fired
        (the D2H read it)."""
        ready = torch.cuda.Event()
        ready.record(stream if stream is not None else self.main)
        return self.down.push(ready, dev_frame[None], [self.out[row]])

    def finish(self):
        self.down.close()
        with _T("main.wait_out_futs"):      # pass-through copies and the first-touch sweep of the output tensor
            for f in self.futs:
                f.result()
        return self.out
