"""IFUNet at 1080p on one MI355X: time per interpolated frame (ensemble on / off) with the kernel split, and --check: parity vs
the oracle at 256x448.  NOT RUN YET — the IFUNet device path was written after round 1's GPU budget was spent."""
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
from cfi_amd.ifunet import IFUNetEngine  # noqa: E402

if __name__ == "__main__":
    sd = synth.ifunet_synth_state_dict(1234)
    eng = IFUNetEngine(sd)
    lib = _lib.load()
    if "--check" in sys.argv:
        from oracle import ifunet_oracle

        fr = synth.smooth_frames(2, 256, 448, seed=2, shift=4.0)
        x = fr.permute(0, 3, 1, 2)
        with torch.inference_mode():
            want = ifunet_oracle.ifunet_forward(sd, x[0:1], x[1:2], 0.5, 1.0, True).permute(0, 2, 3, 1)[0]
        out = torch.empty(256, 448, 3, device="cuda")
        eng.forward(fr[0].cuda().contiguous(), fr[1].cuda().contiguous(), 0.5, out, scale=1.0, ensemble=True)
        d = (out.cpu() - want).abs()
        print(f"IFUNet 256x448 vs oracle: max|d| = {d.max().item():.3e} mean {d.mean().item():.3e}", flush=True)
        eng.release_workspace()
    H, W = 1080, 1920
    fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
    x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
    out = torch.empty(H, W, 3, device="cuda")
    for ens in (True, False):
        for _ in range(2):
            eng.forward(x0, x1, 0.5, out, scale=1.0, ensemble=ens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.forward(x0, x1, 0.5, out, scale=1.0, ensemble=ens)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"IFUNet 1080p ensemble={ens}: {dt * 1e3:.1f} ms/frame = {1 / dt:.1f} frames/s; device memory {torch.cuda.memory_allocated() / 2**30:.2f} GiB", flush=True)
    eng.use_graphs = False      # the event trace needs the launches to pass through the library call by call
    lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
    eng.forward(x0, x1, 0.5, out, scale=1.0, ensemble=True)
    torch.cuda.synchronize()
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    groups = {}
    for k, v in rep.items():
        g = k.split("_")[0] if k.startswith(("conv", "deconv")) else k
        groups[g] = groups.get(g, 0.0) + v[1]
    print("   " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(groups.items(), key=lambda kv: -kv[1])[:14]) + f"  (sum {sum(groups.values()):.1f} ms)")
