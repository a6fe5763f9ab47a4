"""End-to-end node timing (PCIe-inclusive): RIFE_VFI.vfi on a CPU-resident clip, as ComfyUI would call it."""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
ge.load_package()
import cfi_amd.rife as R  # noqa: E402
from cfi_amd import synth  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    sd = synth.rife47_synth_state_dict(1234)
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "rife47.pth")
        torch.save(sd, pth)
        R.load_file_from_github_release = lambda model_type, ckpt: pth
        base = synth.smooth_frames(3, 1080, 1920, seed=1, shift=4.0)
        frames = base[torch.arange(n) % 3].contiguous()
        node = R.RIFE_VFI()
        node.vfi("rife47.pth", frames[:3], multiplier=2, batch_size=bs)  # warm-up (model load, workspace)
        from cfi_amd import _lib
        lib = _lib.load()
        trace = os.environ.get("TRACE", "0") == "1"
        for rep in range(int(os.environ.get('REPS', '4'))):
            if trace:
                lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
            t0 = time.perf_counter()
            res = node.vfi("rife47.pth", frames, multiplier=2, batch_size=bs)
            dt = time.perf_counter() - t0
            if trace:
                lib.vfi_trace_enable(0)
                rep_ = _lib.trace_report()
                print(f"   device kernels: {sum(v[1] for v in rep_.values()):.1f} ms busy of {dt * 1e3:.1f} ms wall; "
                      + ", ".join(f"{k} {v[1]:.1f}" for k, v in sorted(rep_.items(), key=lambda kv: -kv[1][1])[:6]), flush=True)
            out = res[0]    # the previous result is released here, outside the timed region (munmap of 1.6 GB)
            del res
            print(f"node e2e: {n} frames 1080p -> {out.shape[0]} frames, batch_size={bs}: {dt:.3f} s, "
                  f"{(n - 1) / dt:.1f} interpolated frames/s (host tensor in, host tensor out)", flush=True)
            from cfi_amd import hostpipe
            if hostpipe.PROFILE:
                print("   " + "; ".join(f"{k} n={v[0]} {v[1] * 1e3:.0f}ms" for k, v in sorted(hostpipe.stats.items())), flush=True)
                hostpipe.stats.clear()
