"""Launcher for the world_size-2 gloo tests: spawn the ranks, collect rank 0's result, and retry on a fresh port when the
rendezvous itself fails (the free port found a moment ago can be taken by another process before the ranks bind it — seen once
in ~30 runs of the CPU suite).  A wrong RESULT is never retried: only a rank that died or a result that never arrived."""
import queue
import socket

import torch.multiprocessing as mp


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(worker, world=2, timeout=300, attempts=3):
    ctx = mp.get_context("spawn")
    last = None
    for _ in range(attempts):
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = None
        try:
            got = q.get(timeout=timeout)
        except queue.Empty:
            last = "no result from rank 0"
        except Exception as e:      # a result that could not be received is a transport failure, not a wrong result: retry
            last = f"receiving rank 0's result failed: {type(e).__name__}: {e}"
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.kill()          # the exact process this test started
                p.join()
        codes = [p.exitcode for p in procs]
        if got is not None and all(c == 0 for c in codes):
            import torch

            return torch.from_numpy(got) if not isinstance(got, torch.Tensor) else got
        last = f"{last or 'rank failure'}; exit codes {codes}"
    raise AssertionError(f"{attempts} attempts failed: {last}")
