"""RIFE arch 4.0 (op-by-op engine) at 1080p on one MI355X: default widgets (fast path) and with the Unet refinement."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
ge.load_package()
from cfi_amd import _lib, synth  # noqa: E402
from cfi_amd.rife40 import Rife40Engine  # noqa: E402

if __name__ == "__main__":
    B, H, W = 4, 1080, 1920
    fr = synth.smooth_frames(3, H, W, seed=1, shift=4.0)
    dev = [fr[i % 3].cuda().contiguous() for i in range(B + 1)]
    out = torch.empty((B, H, W, 3), device="cuda")
    eng = Rife40Engine(synth.rife40_synth_state_dict(1234))
    eng.configure(H, W, B)
    for training, fastmode, label in ((True, True, "fast_mode=True ensemble=True (node defaults: blocks only)"),
                                      (False, False, "fast_mode=False ensemble=False (+ scale test + Contextnet/Unet refinement)")):
        def step():
            eng.forward(dev[:B], dev[1:B + 1], [0.5] * B, [8.0, 4.0, 2.0, 1.0], training, fastmode, out)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        print(f"RIFE 4.0 {label}: {dt * 1e3:.2f} ms per step of {B} -> {B / dt:.1f} interpolated 1080p frames/s", flush=True)
        lib = _lib.load()
        lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
        step()
        torch.cuda.synchronize()
        lib.vfi_trace_enable(0)
        rep = _lib.trace_report()
        tot = sum(v[1] for v in rep.values())
        print("   " + ", ".join(f"{k} {v[1]:.2f}" for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:10]) + f"  (sum {tot:.2f} ms)")
        lib.vfi_trace_reset()
    eng.close()
