"""GMFSS Fortuna (union) at 1080p on one MI355X, first-correct path: time of prepare() (Model.reuse: FeatureNet, GMFlow both
directions, MetricNet) and render(t) (Model.inference), with the kernel split."""
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
from cfi_amd.gmfss import GMFSSEngine  # noqa: E402

def _rate(k, v):
    """VFI_TRACE_SHAPES=1: TFLOP/s of a conv / deconv row named conv3x3s1_64to64@1x544x960"""
    import re
    m = re.match(r"(conv|deconv)(\d)x(\d)s(\d)_(\d+)to(\d+)@(\d+)x(\d+)x(\d+)", k)
    if not m:
        return ""
    taps = int(m[2]) * int(m[3]) / (4 if m[1] == "deconv" else 1)
    return f" [{2 * taps * int(m[5]) * int(m[6]) * int(m[7]) * int(m[8]) * int(m[9]) * v[0] / v[1] / 1e9:.0f} TF]"


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    H, W = (int(args[0]), int(args[1])) if len(args) >= 2 else (1080, 1920)
    if "--coherent" in sys.argv:      # matched weights + textured frames: GMFlow finds the true (small) motion, as a trained model does
        eng = GMFSSEngine(synth.gmfss_coherent_state_dicts(3, "union"))
        fr = synth.texture_frames(2, H, W, seed=5)
    else:                             # random weights: the flow is noise of arbitrary magnitude (worst case for the splats)
        eng = GMFSSEngine(synth.gmfss_synth_state_dicts(1234))
        fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
    x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
    out = torch.empty(H, W, 3, device="cuda")
    lib = _lib.load()
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.prepare(x0, x1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.render(0.5, out)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"GMFSS union {H}x{W} (rep {rep}): prepare {1e3 * (t1 - t0):.1f} ms/pair, render {1e3 * (t2 - t1):.1f} ms/frame; "
              f"device memory {torch.cuda.memory_allocated() / 2**30:.2f} GiB", flush=True)
    fl = eng.prepared["flows"] if isinstance(getattr(eng, "prepared", None), dict) and "flows" in eng.prepared else None
    if fl is not None:
        a = fl.abs().flatten()
        print(f"   flow |f|: mean {a.mean().item():.2f}, p99 {a.kthvalue(int(0.99 * a.numel())).values.item():.2f}, max {a.max().item():.2f} px", flush=True)
    print(f"   workspace: {eng.workspace_bytes() / 2**30:.2f} GiB (pooled scratch: {len(eng._pool.chunks)} chunks)", flush=True)
    eng.use_graphs = False      # the event trace needs the launches to pass through the library call by call
    for phase in ("prepare", "render"):        # per-kernel split of each phase
        lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
        if phase == "prepare":
            eng.prepare(x0, x1)
        else:
            eng.render(0.5, out)
        torch.cuda.synchronize()
        lib.vfi_trace_enable(0)
        rep = _lib.trace_report()
        print(f"   {phase}: " + ", ".join(f"{k} {v[0]}x {v[1]:.2f}{_rate(k, v)}" for k, v in sorted(rep.items(), key=lambda kv: -kv[1][1])[:22])
              + f"  (sum {sum(v[1] for v in rep.values()):.1f} ms)", flush=True)
    lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
    eng.prepare(x0, x1)
    eng.render(0.5, out)
    torch.cuda.synchronize()
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    groups = {}
    for k, v in rep.items():
        g = k.split("_")[0] if k.startswith(("conv", "deconv")) else k
        groups[g] = groups.get(g, 0.0) + v[1]
    tot = sum(groups.values())
    print("   " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(groups.items(), key=lambda kv: -kv[1])[:14]) + f"  (sum {tot:.1f} ms)", flush=True)
    print(f"   output finite: {bool(torch.isfinite(out).all())}, range [{out.min().item():.3f}, {out.max().item():.3f}]")
