"""M2M at 1080p on one MI355X (BASELINE.json configs[3]): parity vs the oracle at full size + timing + kernel split.
prepare() = everything timestep independent (flow network + motion refinement), render(t) = one splat."""
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
from cfi_amd.m2m import M2MEngine  # noqa: E402


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


if __name__ == "__main__":
    H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 and sys.argv[1].isdigit() else (1080, 1920)
    check = "--check" in sys.argv
    sd = synth.m2m_synth_state_dict(1234)
    eng = M2MEngine(sd)
    fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
    x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
    out = eng.forward(x0, x1, 0.5)
    torch.cuda.synchronize()
    print(f"M2M {H}x{W}: device memory after first forward {torch.cuda.memory_allocated() / 2**30:.2f} GiB (torch buffers)", flush=True)
    for rep in range(2):
        tp = timed(lambda: eng.prepare(x0, x1), 5)
        tr = timed(lambda: eng.render(0.5), 10)
        print(f"M2M {H}x{W}: prepare {tp * 1e3:.2f} ms/pair, render {tr * 1e3:.2f} ms/frame -> 2x: {1 / (tp + tr):.1f} frames/s, "
              f"8x: {7 / (tp + 7 * tr):.1f} frames/s (reference semantics re-run the network per frame: {1 / (tp + tr):.1f})", flush=True)
    lib = _lib.load()
    lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
    eng.forward(x0, x1, 0.5)
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    tot = sum(v[1] for v in rep.values())
    print(f"traced total {tot:.2f} ms")
    import re
    for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
        m = re.match(r"(conv|deconv)(\d)x(\d)s(\d)_(\d+)to(\d+)@(\d+)x(\d+)x(\d+)", k)      # VFI_TRACE_SHAPES=1: rate per shape
        rate = ""
        if m:
            taps = int(m[2]) * int(m[3]) / (4 if m[1] == "deconv" else 1)      # a transposed 4x4 s2 conv: 4 taps per output pixel
            rate = f"  {2 * taps * int(m[5]) * int(m[6]) * int(m[7]) * int(m[8]) * int(m[9]) * v[0] / v[1] / 1e9:6.1f} TFLOP/s"
        print(f"   {k:40s} {v[0]:4d} calls {v[1]:9.3f} ms {100 * v[1] / tot:5.1f}%{rate}")
    if check:
        from oracle import m2m_model_oracle as mo
        t0 = time.time()
        x = fr.permute(0, 3, 1, 2)
        with torch.inference_mode():
            want = mo.m2m_forward(sd, x[0:1], x[1:2], [torch.tensor([0.5]).view(1, 1, 1, 1)])[0]
        d = (out.cpu() - want[0].permute(1, 2, 0)).abs()
        print(f"M2M {H}x{W} vs oracle: max|d| = {d.max().item():.3e} mean {d.mean().item():.3e} (oracle CPU {time.time() - t0:.1f}s), "
              f"output range [{want.min().item():.2f},{want.max().item():.2f}]", flush=True)
