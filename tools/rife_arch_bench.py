"""Device-side throughput of the RIFE arch variants at 1080p (frames resident in HBM, batch 8), same procedure as bench.py's
timed region: per step B new frames are prepared+encoded and B pairs interpolated."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
ge.load_package()
from cfi_amd import synth  # noqa: E402
from cfi_amd.rife import RifeEngine  # noqa: E402

if __name__ == "__main__":
    B, H, W = 8, 1080, 1920
    fr = synth.smooth_frames(3, H, W, seed=1, shift=4.0)
    dev = [fr[i % 3].cuda().contiguous() for i in range(B + 1)]
    out = torch.empty((B, H, W, 3), device="cuda")
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--arch=")]
    for arch, sdf in (("4.7", synth.rife47_synth_state_dict), ("4.17", synth.rife417_synth_state_dict), ("4.26", synth.rife426_synth_state_dict)):
        if only and arch not in only:
            continue
        eng = RifeEngine(sdf(1234), arch)
        eng.configure(H, W, B, 2 * B + 2, 1.0)
        for i in range(B + 1):
            eng.load_frame(i, dev[i])

        def step():
            for i in range(B):     # B new frames per step, like a stream
                eng.load_frame(i, dev[i])
            eng.interpolate(list(range(B)), list(range(1, B + 1)), [0.5] * B, out)

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        flop, _ = eng.work_per_task()
        print(f"RIFE {arch}: {dt * 1e3:.2f} ms per step of {B} -> {B / dt:.1f} interpolated 1080p frames/s, "
              f"{flop * B / dt / 1e12:.1f} TFLOP/s over the whole network ({flop / 1e9:.1f} GFLOP/frame)", flush=True)
        if "--split" in sys.argv:
            from cfi_amd import _lib
            lib = _lib.load()
            lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
            step()
            torch.cuda.synchronize()
            lib.vfi_trace_enable(0)
            rep = _lib.trace_report()
            tot = sum(v[1] for v in rep.values())
            print("   " + ", ".join(f"{k} {v[1]:.2f}" for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])) + f"  (sum {tot:.2f} ms)")
            lib.vfi_trace_reset()
        eng.close()
