"""IFRNet_L / IFRNet_S at 1080p on one MI355X: timing + kernel split (+ --check: parity vs the oracle at full size).
The node's default call (multiplier 2) runs the network at working resolution 0.5 with time embedding 1.0 (the
reference's positional mis-binding, see ifrnet.py); 1.0/0.5 = the network as its authors call it."""
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
from cfi_amd.ifrnet import IFRNetEngine  # noqa: E402

if __name__ == "__main__":
    H, W = 1080, 1920
    check = "--check" in sys.argv
    fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
    x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
    out = torch.empty(1, H, W, 3, device="cuda")
    lib = _lib.load()
    for kind in ("L", "S"):
        sd = synth.ifrnet_synth_state_dict(kind, 1234)
        eng = IFRNetEngine(sd, kind)
        for sf, t in ((0.5, 1.0), (1.0, 0.5)):
            for _ in range(2):
                eng.forward([x0], [x1], sf, t, out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                eng.forward([x0], [x1], sf, t, out)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"IFRNet_{kind} 1080p working resolution x{sf}: {dt * 1e3:.2f} ms/frame = {1 / dt:.1f} frames/s", flush=True)
            lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
            eng.forward([x0], [x1], sf, t, out)
            torch.cuda.synchronize()
            lib.vfi_trace_enable(0)
            rep = _lib.trace_report()
            tot = sum(v[1] for v in rep.values())
            groups = {}
            for k, v in rep.items():
                g = k.split("_")[0] if k.startswith(("conv", "deconv")) else k
                groups[g] = groups.get(g, 0.0) + v[1]
            print("   " + ", ".join(f"{k} {v:.2f}" for k, v in sorted(groups.items(), key=lambda kv: -kv[1])) + f"  (sum {tot:.2f} ms)", flush=True)
            lib.vfi_trace_reset()
            if check and sf == 0.5:
                from oracle import ifrnet_oracle
                x = fr.permute(0, 3, 1, 2)
                t1 = time.time()
                with torch.inference_mode():
                    want = ifrnet_oracle.ifrnet_forward(sd, x[0:1], x[1:2], sf, t).permute(0, 2, 3, 1)
                d = (out.cpu() - want).abs()
                print(f"   vs oracle at 1080p: max|d| = {d.max().item():.3e} mean {d.mean().item():.3e} (oracle CPU {time.time() - t1:.1f}s)", flush=True)
        eng.close()
