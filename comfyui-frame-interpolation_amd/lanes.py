"""Pair lanes: several frame pairs of a clip in flight on one GPU, each on its own HIP stream with its own engine.

The generic node loop of the reference (vfi_utils.py:149-389, film/__init__.py:63-113) walks the clip pair by pair; the pairs are
independent.  One pair of M2M / FILM / GMFSS / IFUNet / IFRNet at 1080p is a chain of 100-1500 launches, many of them on coarse
pyramid levels that fill a fraction of the 256 compute units (4-60 workgroups) or sit in their own pipeline latency.  A second
and third pair on other streams fill those holes: same kernels, same launch geometry, bit-identical frames, 1.06x (FILM) to
1.19x (IFUNet, M2M) the frames/s of one stream (tools/overlap_probe.py, profiles/r06_pair_lanes.txt).  The price is one
workspace per lane — sized for 288 GB of HBM, not for 24.

Everything a lane shares with its siblings is read-only (packed weights are per engine; the library's scratch — split-K sums,
attention partials, splat lists — is keyed by (device, stream)).
"""
import os

import torch

# lanes per model type (VFI_PAIR_LANES=<n> overrides all of them; 1 = the single-stream loop)
DEFAULT_LANES = {"m2m": 3, "film": 2, "gmfss": 3, "ifunet": 3, "ifrnet": 2}
# A lane's first pair pays for its workspace (and, GMFSS / IFUNet, for its graph capture; FILM's first forward allocates its scratch
# call by call and holds the host for a whole pair): a lane is only opened for this many pairs of the clip (measured at 1080p:
# FILM with 2 lanes on 12 pairs 583 vs 567 ms and on 24 pairs 1128 vs 1122-1131 ms (the node releases and re-allocates every workspace per call), M2M with 3 lanes on 8 pairs 89 vs 76 ms — tools/node_e2e_models.py)
# GMFSS / IFUNet: a lane's first two pairs cost 150 / 130 ms more than steady ones (workspace fill, graph capture —
# tools/engine_build_probe.py) against 5.5 / 5.3 ms gained per pair of the clip; their workspaces and graphs now stay between calls of one
# frame shape (ckpt.end_call), so only a process's first call of a shape pays: a lane per 8 pairs.
# M2M / IFRNet: a lane's workspace costs ~2 ms of device time and ~6 ms of host time per call: three lanes on an 8-pair clip lost (89 vs 76 ms),
# on a 12-pair clip they gained 6 %.
PAIRS_PER_LANE = {"film": 24, "gmfss": 8, "ifunet": 8, "m2m": 4, "ifrnet": 4}


def lanes_for(model):
    env = os.environ.get("VFI_PAIR_LANES")
    if env:
        return max(1, int(env))
    return DEFAULT_LANES.get(model, 1)


def lane_set(model, build):
    """the node classes' LaneSet of a model type"""
    return LaneSet(build, lanes_for(model), pairs_per_lane=PAIRS_PER_LANE.get(model, 2))


class LaneSet:
    """K engines built from one checkpoint by ``build()``; engine i runs on stream i.  Engines beyond the first are built at first
    use (a two-frame clip never pays for them).  Looks like an engine to the node code that owns it: ``device``, ``close()``,
    ``release_workspace()``; attributes set through ``configure`` reach every lane."""

    def __init__(self, build, k, first=None, pairs_per_lane=2):
        """first: an engine the caller already owns becomes lane 0 (close() leaves it alone)"""
        self._build, self.k, self.pairs_per_lane = build, max(1, int(k)), max(1, int(pairs_per_lane))
        self._borrowed = first is not None
        self.engines = [first if first is not None else build()]
        self.device = self.engines[0].device
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)      # whatever the constructor queued (fills, uploads) is done before another stream reads it
        self._streams = []
        self._setup = {}           # key -> fn(engine): applied to every lane, present and future (configure)
        self.apart_from = []      # torch streams that are busy beside the lanes (the node loops put their copy streams here)

    def configure(self, fn, key="node"):
        """fn(engine): per-call settings (the node's: IFRNet's embt, IFUNet's scale / ensemble; the loop's: lone_pair) on every lane,
        present and future.  A later call with the same key replaces the earlier one."""
        self._setup[key] = fn
        for e in self.engines:
            fn(e)

    def lane(self, i):
        """-> (engine, stream) of lane i < k"""
        from . import _lib

        if not self._streams:
            # not torch.cuda.Stream(): its pool of 32 hands one stream to two owners.  All k at once, probed pairwise: streams that were
            # bound to one hardware queue run strictly in turn (two such lanes = one), and a lane behind the copy-back stream's event
            # waits stalls with it — `apart_from` lists the busy streams the lanes should stay clear of where the queues allow
            self._streams = _lib.own_streams_apart(self.device, self.k, avoid=self.apart_from)
        while len(self.engines) <= i:
            with torch.cuda.stream(self._streams[len(self.engines)].stream):      # the constructor's device work is ordered with the lane's first pair
                e = self._build()
                for fn in self._setup.values():
                    fn(e)
            self.engines.append(e)
        return self.engines[i], self._streams[i].stream

    def release_workspace(self):
        for e in self.engines:
            e.release_workspace()

    def workspace_bytes(self):
        """device bytes the lanes' workspaces hold, or None when an engine cannot say (ckpt.end_call then releases as before)"""
        if not all(hasattr(e, "workspace_bytes") for e in self.engines):
            return None
        return sum(e.workspace_bytes() for e in self.engines)

    def close(self):
        if self._streams and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        for e in self.engines[1 if self._borrowed else 0:]:
            e.close()
        self.engines = self.engines[:0]
        for st in self._streams:
            st.release()
        self._streams = []


def configure(engine, fn, key="node"):
    """fn(engine) on a plain engine, on every lane of a LaneSet"""
    if isinstance(engine, LaneSet):
        engine.configure(fn, key)
    else:
        fn(engine)


def tell_lone_pair(engine, n_lanes):
    """Engines that fork one pair's independent stages onto a side stream of their own (FILM's two image / flow halves, IFUNet's
    ensemble and ResynNet passes) do so only when theirs is the only pair in flight: with pair lanes open the device's four
    hardware queues are taken (FILM: two lanes 42.0 ms per pair without the inner fork, 42.9 with it)."""
    one = n_lanes == 1
    configure(engine, lambda e: e.lone_pair(one) if hasattr(e, "lone_pair") else None, key="lone_pair")


def lanes_of(engine, n_pairs):
    """(lane getter, lane count) for a node loop over n_pairs pairs: a LaneSet spreads them, a plain engine is one lane on the
    current stream."""
    if isinstance(engine, LaneSet):
        n = min(engine.k, n_pairs // engine.pairs_per_lane)
        if n > 1:
            return engine.lane, n
    eng = engine.engines[0] if isinstance(engine, LaneSet) else engine
    main = torch.cuda.current_stream(eng.device) if eng.device.type == "cuda" else None
    return (lambda i: (eng, main)), 1
