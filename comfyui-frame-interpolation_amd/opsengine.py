"""Shared plumbing of the op-by-op engines (GMFSS Fortuna, IFUNet): a backend (libvfi_hip.so on the current GPU — or, in the CPU
orchestration tests only, the test double handed in through ``_test_backend``), scratch tensors, vfi_conv layer objects built
from checkpoint tensors, and thin wrappers around the generic C-ABI calls.  Every ``torch.cat`` of the reference becomes a
channel window of a pre-allocated NHWC tensor: wrappers take (tensor, channel offset) pairs."""
import contextlib
import ctypes as C

import torch

from . import _lib


def _p(t, off=0):
    return t.data_ptr() + 4 * off


def _cs(c):
    return (c + 7) // 8 * 8


class _Device:
    """The real backend: libvfi_hip.so on the current GPU (there is no CPU fallback)."""

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("VFI (HIP): no GPU visible; this node has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _lib.check(self.lib.vfi_init(self.device.index or 0), "vfi_init")

    @staticmethod
    def stream():
        return _lib.stream_ptr()

    @staticmethod
    def last_error():
        return _lib.last_error()


class _Pool:
    """Workspace memory of a pooled engine: a few large, zero-initialised device buffers with a first-fit free list each (blocks
    split on allocation, merged with their neighbours on release).  The sequence of requests of an engine is the same on every
    call, hence so are the addresses; a recycled block holds the finite activations of its previous user (channel padding that
    a convolution reads against zero weights stays harmless)."""
    CHUNK = 512 << 20

    def __init__(self, device):
        self.device, self.chunks, self.free = device, [], []      # free[i]: sorted [offset, size] holes of chunk i

    def _new_chunk(self, nbytes):
        return torch.zeros(nbytes // 4, dtype=torch.float32, device=self.device)

    def take(self, nbytes):
        n = max(256, (int(nbytes) + 255) // 256 * 256)
        for i, holes in enumerate(self.free):
            for k, (off, size) in enumerate(holes):
                if size >= n:
                    if size == n:
                        holes.pop(k)
                    else:
                        holes[k] = [off + n, size - n]
                    return (i, off, n)
        size = max(self.CHUNK, n)
        self.chunks.append(self._new_chunk(size))
        self.free.append([[n, size - n]] if size > n else [])
        return (len(self.chunks) - 1, 0, n)

    def give(self, blk):
        i, off, n = blk
        holes = self.free[i]
        k = 0
        while k < len(holes) and holes[k][0] < off:
            k += 1
        holes.insert(k, [off, n])
        if k + 1 < len(holes) and holes[k][0] + holes[k][1] == holes[k + 1][0]:      # merge with the hole after
            holes[k][1] += holes.pop(k + 1)[1]
        if k > 0 and holes[k - 1][0] + holes[k - 1][1] == holes[k][0]:               # ... and the one before
            holes[k - 1][1] += holes.pop(k)[1]

    def view(self, blk, shape):
        numel = 1
        for d in shape:
            numel *= d
        return self.chunks[blk[0]][blk[1] // 4:blk[1] // 4 + numel].view(shape)

    def nbytes(self):
        return sum(b.numel() * 4 for b in self.chunks)


class OpsEngine:
    # r6: a call whose launch sequence is fixed (same shapes, same scalars) is captured ONCE into a HIP graph and replayed: GMFSS issues
    # ~1500 library calls per pair and IFUNet ~2000 through ctypes — 28-29 ms of interpreter time per pair beside 33-43 ms of device
    # time, which left the node loop host-bound as soon as a second pair lane wanted feeding (lanes.py).  Tools that read the
    # library's event trace or count FLOP per layer set ``use_graphs = False`` (a replay does not pass through Python).
    use_graphs = True
    MAX_GRAPHS = 64

    def __init__(self, device=None, _test_backend=None, pooled=False):
        self.be = _test_backend if _test_backend is not None else _Device(device)
        self.lib, self.device = self.be.lib, self.be.device
        self.handles, self.scratch, self.consts = [], {}, {}
        # pooled engines: scratch tensors come from a _Pool and die with the scope (``with self._scope():``) they were first asked
        # for in, or when dropped explicitly; un-pooled engines keep every named tensor for the life of the workspace
        self._pool = _Pool(self.device) if pooled else None
        self._live = [{}]
        self._held = None
        self.conv_flop = None      # set to 0 to count the direct-form FLOP of every convolution launched from here on (bench.py other_paths)
        self._graphs, self._cap_stream = {}, None
        self._sides = {}           # hipStream_t of a caller's stream -> the OwnStream this engine forks onto beside it (_fork)

    # ---- HIP graphs -------------------------------------------------------------------------------------------------
    def _replayable(self, key, ins, outs, fn):
        """fn(*ins, *outs): a fixed sequence of library launches on the engine's own (address-stable) workspace, reading the device
        tensors ``ins`` and writing ``outs``.  First call of a ``key`` (shapes and scalars of the call): eager, on the caller's stream.
        Second call: the sequence runs once on the engine's capture stream — every lazily grown buffer behind THAT stream (pool
        chunks, the library's per-stream scratch) exists afterwards — and is then captured there with static copies of ins / outs; that
        call and all later ones replay the graph (ins copied in, outs copied out: 3 x 25 MB of device copies beside tens of ms of kernels).  Anything that makes a
        capture impossible leaves the key on the eager path, loudly."""
        g = self._graphs.get(key) if self.use_graphs and self.conv_flop is None and self.device.type == "cuda" and isinstance(self.be, _Device) else False
        if g is False:
            return fn(*ins, *outs)
        cur = torch.cuda.current_stream(self.device)
        if g is None or g == "seen":
            # A key is captured at its SECOND use: a call shape that occurs once (a 2-frame clip, one of 999 timesteps of a x1000 pair)
            # must not pay for a warm-up run and a capture it will never replay; and an engine keeps at most MAX_GRAPHS of them.
            if g is None or sum(1 for v in self._graphs.values() if isinstance(v, tuple)) >= self.MAX_GRAPHS:
                self._graphs[key] = "seen"
                return fn(*ins, *outs)
            if self._cap_stream is None:
                # the engine's own stream (never one of torch's pooled 32): the library scratch behind (device, stream) is then this
                # engine's alone, and the addresses a graph bakes in stay valid for as long as the engine keeps the stream
                self._cap_stream = _lib.OwnStream(self.device)
            cs = self._cap_stream.stream
            sin, sout = [torch.empty_like(x) for x in ins], [torch.empty_like(x) for x in outs]
            try:
                cs.wait_stream(cur)
                with torch.cuda.stream(cs):
                    for a, b in zip(sin, ins):
                        a.copy_(b, non_blocking=True)
                    fn(*sin, *sout)                       # warm-up: grows what has to grow, on the stream the capture will run on
                graph = torch.cuda.CUDAGraph()
                # thread_local: the host pipeline's worker threads (event waits, copies on their own streams) keep running meanwhile
                with torch.cuda.graph(graph, stream=cs, capture_error_mode="thread_local"):
                    ret = fn(*sin, *sout)                 # (its Python-side result — GMFSS' prepared-state record — belongs to the graph)
                g = self._graphs[key] = (graph, sin, sout, ret)
                cur.wait_stream(cs)
            except Exception as e:  # noqa: BLE001 — keep the engine usable: this key stays eager
                import warnings

                self._graphs[key] = False
                torch.cuda.synchronize(self.device)
                warnings.warn(f"{type(self).__name__}: HIP graph capture of {key} failed ({type(e).__name__}: {e}); this call shape runs eagerly",
                              RuntimeWarning, stacklevel=2)
                return fn(*ins, *outs)
        graph, sin, sout, ret = g
        for a, b in zip(sin, ins):
            a.copy_(b, non_blocking=True)
        graph.replay()
        for a, b in zip(outs, sout):
            a.copy_(b, non_blocking=True)
        return ret

    # ---- two independent stages side by side ----------------------------------------------------------------------------
    def _fork(self):
        """-> (current stream, side stream) with the side stream ordered after everything queued on the current one, or (None, None) where
        there is nothing to fork onto (the CPU test double).  The side stream is the engine's own, probed onto ANOTHER hardware queue
        than the current stream (two streams of one queue run in turn: _lib.streams_share_queue) — once per current stream: the eager
        first call and the warm-up run in front of a graph capture do the probing, the capture itself finds the choice made."""
        if self.device.type != "cuda" or not isinstance(self.be, _Device):
            return None, None
        cur = torch.cuda.current_stream(self.device)
        own = self._sides.get(cur.cuda_stream)
        if own is None:
            own = self._sides[cur.cuda_stream] = _lib.own_streams_apart(self.device, 1, avoid=[cur])[0]
        own.stream.wait_stream(cur)
        return cur, own.stream

    @staticmethod
    def _join(cur, side):
        if cur is not None:
            cur.wait_stream(side)

    def _drop_graphs(self):
        if self._graphs and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)      # a replay may still be in flight
        self._graphs = {}

    # ---- plumbing ---------------------------------------------------------------------------------------------------
    def _c(self, name, *args):
        rc = getattr(self.lib, name)(*args, self.be.stream())
        if rc != 0:
            raise RuntimeError(f"{name} failed (code {rc}): {self.be.last_error()}")

    def _t(self, name, *shape):
        key = (name,) + tuple(shape)
        if self._pool is None:
            if key not in self.scratch:
                self.scratch[key] = torch.zeros(shape, dtype=torch.float32, device=self.device)
            return self.scratch[key]
        for d in reversed(self._live):
            if key in d:
                return d[key][0]
        numel = 1
        for n in shape:
            numel *= n
        blk = self._pool.take(numel * 4)
        t = self._pool.view(blk, shape)
        self._live[-1][key] = (t, blk)
        return t

    @contextlib.contextmanager
    def _scope(self):
        """scratch tensors first requested inside the block go back to the pool when it ends (no-op for un-pooled engines)"""
        self._live.append({})
        try:
            yield
        finally:
            for _, blk in self._live.pop().values():
                if self._pool is None:
                    continue
                if self._held is not None and self._held["active"]:
                    self._held["blocks"].append(blk)      # a forked stage's temporaries: kept until the join (_hold)
                else:
                    self._pool.give(blk)

    @contextlib.contextmanager
    def _hold(self):
        """While ``h["active"]`` is set inside the block, scopes that end keep their pool blocks; all of them go back when the block
        ends.  For a stage forked onto the side stream: the stage on the current stream must not be handed the forked stage's
        temporaries while the side stream still works on them (the pool recycles by program order, i.e. for ONE stream)."""
        assert self._held is None, "holds do not nest"
        self._held = h = {"active": False, "blocks": []}
        try:
            yield h
        finally:
            self._held = None
            if self._pool is not None:
                for blk in h["blocks"]:
                    self._pool.give(blk)

    def _drop(self, *tensors):
        """give scratch tensors back before their scope ends (GridNet's states, dead long before the network is done)"""
        if self._pool is None:
            return
        ids = {id(t) for t in tensors}            # the very objects _t returned (not views of them)
        for d in self._live:
            for key in [k for k, (t, _) in d.items() if id(t) in ids]:
                self._pool.give(d.pop(key)[1])

    def workspace_bytes(self):
        legacy = sum(t.numel() * t.element_size() for t in self.scratch.values() if torch.is_tensor(t))
        return legacy + (self._pool.nbytes() if self._pool is not None else 0)

    def _const(self, key, make):
        if key not in self.consts:
            t = make()
            self.consts[key] = t.to(self.device, torch.int32 if not t.is_floating_point() else torch.float32).contiguous()
        return self.consts[key]

    def _layer(self, w, b=None, slopes=None, kind=0, stride=1, chan_map=None, cin_phys=None, scale_out=None, shift_out=None):
        """vfi_conv layer object from checkpoint tensors.  w: Conv2d [co,ci,k,k] / Linear [co,ci] / ConvTranspose2d [ci,co,4,4];
        slopes: scalar PReLU slope (replicated) or per-channel vector; scale_out: per-output-channel factor folded into the
        weights and bias (ResConv's beta; an eval-mode BatchNorm's gamma / sqrt(var + eps)); shift_out: added to the bias afterwards
        (BatchNorm's beta - mean * scale)."""
        w = w.detach().to("cpu", torch.float32)
        if w.dim() == 2:
            w = w[:, :, None, None]
        cout, cin = (w.shape[0], w.shape[1]) if kind == 0 else (w.shape[1], w.shape[0])
        b = torch.zeros(cout) if b is None else b.detach().to("cpu", torch.float32)
        if scale_out is not None:
            s = scale_out.detach().to("cpu", torch.float32).reshape(-1)
            w = w * (s.view(-1, 1, 1, 1) if kind == 0 else s.view(1, -1, 1, 1))
            b = b * s
        if shift_out is not None:
            b = b + shift_out.detach().to("cpu", torch.float32).reshape(-1)
        w, b = w.contiguous(), b.contiguous()
        pr = None
        if slopes is not None:
            pr = slopes.detach().to("cpu", torch.float32).reshape(-1)
            pr = (pr.repeat(cout) if pr.numel() == 1 else pr).contiguous()
        cin_phys = cin_phys or _cs(cin)
        cm = (C.c_int * cin)(*chan_map) if chan_map is not None else None
        h = self.lib.vfi_conv_create_ex(kind, w.data_ptr(), b.data_ptr(), cout, cin, w.shape[2], stride, 0, cm, cin_phys,
                                        pr.data_ptr() if pr is not None else None)
        if not h:
            raise RuntimeError("vfi_conv_create_ex failed: " + self.be.last_error())
        self.handles.append(h)
        return dict(h=h, kind=kind, stride=stride, cout=cout, cin=cin, k=int(w.shape[2]), act=3 if pr is not None else 0)

    def _conv(self, L, src, soff, dst, doff, act=None, slope=0.0, res=None):
        n, hin, win, cs = src.shape
        want = (hin * 2, win * 2) if L["kind"] == 1 else (hin // L["stride"], win // L["stride"])
        assert tuple(dst.shape[1:3]) == want and dst.shape[0] == n, (src.shape, dst.shape, want)
        if self.conv_flop is not None:      # direct form: 2 x taps x Cin x Cout per output pixel (a ConvTranspose2d(4, 2, 1) has 4 taps per output pixel)
            self.conv_flop += 2.0 * (4 if L["kind"] == 1 else L["k"] * L["k"]) * L["cin"] * L["cout"] * n * want[0] * want[1]
        self._c("vfi_conv_forward_ex", L["h"], _p(src, soff), cs, hin, win, _p(dst, doff), dst.shape[-1], n,
                L["act"] if act is None else act, slope, 0.0, 0.0, _p(res) if res is not None else None,
                res.shape[-1] if res is not None else 0)

    def _ax(self, a, aoff, b, boff, out, ooff, c, alpha=1.0, beta=1.0, px=None):
        px = px if px is not None else a.shape[0] * a.shape[1] * a.shape[2]
        self._c("vfi_axpby", _p(a, aoff), a.shape[-1], _p(b, boff) if b is not None else None, b.shape[-1] if b is not None else 0,
                _p(out, ooff), out.shape[-1], px, c, alpha, beta)

    def _resize(self, src, soff, dst, doff, c, mul=1.0):
        self._c("vfi_resize_bilinear", _p(src, soff), src.shape[-1], _p(dst, doff), dst.shape[-1], src.shape[0], src.shape[1], src.shape[2],
                dst.shape[1], dst.shape[2], c, mul)

    def close(self):
        for h in self.handles:
            self.lib.vfi_conv_destroy(h)
        self.handles = []
        self.release_workspace()
        if getattr(self, "_cap_stream", None) is not None:
            self._cap_stream.release()
            self._cap_stream = None
        for own in getattr(self, "_sides", {}).values():
            own.release()
        self._sides = {}

    def release_workspace(self):
        self._drop_graphs()        # they hold the addresses of this workspace
        self.scratch = {}
        if self._pool is not None:
            self._pool, self._live = _Pool(self.device), [{}]

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001 — interpreter shutdown
            pass

    def _prelu(self, src, soff, dst, doff, c, slope):
        self._c("vfi_prelu_scalar", _p(src, soff), src.shape[-1], _p(dst, doff), dst.shape[-1], c, src.shape[0] * src.shape[1] * src.shape[2], slope)
