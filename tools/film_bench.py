"""FILM 2x at 1080p on one MI355X (BASELINE.json configs[2]): parity vs the oracle at full size + timing + kernel split."""
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
from cfi_amd.film import FilmEngine  # noqa: E402

if __name__ == "__main__":
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
    check = "--check" in sys.argv
    sd = synth.film_synth_state_dict(1234)
    eng = FilmEngine(sd)
    fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
    x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
    out = eng.forward(x0, x1)
    torch.cuda.synchronize()
    print(f"FILM {H}x{W}: device memory after first forward {torch.cuda.memory_allocated() / 2**30:.1f} GiB (torch buffers)", flush=True)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = eng.forward(x0, x1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"FILM {H}x{W}: {dt * 1e3:.2f} ms per interpolated frame = {1 / dt:.2f} frames/s; "
              f"{8823.8 * (H * W) / (1080 * 1920) / dt / 1e3:.1f} TFLOP/s on 8.82 TFLOP/frame@1080p", flush=True)
    lib = _lib.load()
    lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
    eng.forward(x0, x1)
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    tot = sum(v[1] for v in rep.values())
    import re
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
        m = re.match(r"conv(\d)x(\d)_(\d+)to(\d+)@(\d+)x(\d+)", k)      # VFI_TRACE_SHAPES=1: per-shape rows with their rate
        rate = f"  {2 * int(m[1]) * int(m[2]) * int(m[3]) * int(m[4]) * int(m[5]) * int(m[6]) * v[0] / v[1] / 1e9:6.1f} TFLOP/s" if m else ""
        print(f"   {k:30s} {v[0]:4d} calls {v[1]:9.3f} ms {100 * v[1] / tot:5.1f}%{rate}")
    if check:
        from oracle import film_oracle
        t0 = time.time()
        with torch.inference_mode():
            want = film_oracle.film_forward(sd, fr[0:1].permute(0, 3, 1, 2).contiguous(), fr[1:2].permute(0, 3, 1, 2).contiguous())
        d = (out.cpu() - want[0].permute(1, 2, 0)).abs()
        print(f"FILM {H}x{W} vs oracle: max|d| = {d.max().item():.3e} mean {d.mean().item():.3e} (oracle CPU {time.time() - t0:.1f}s), "
              f"output range [{want.min().item():.2f},{want.max().item():.2f}]", flush=True)
