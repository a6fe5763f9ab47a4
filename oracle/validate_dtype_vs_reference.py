"""ORACLE tooling — what does the reference's ``dtype`` widget (float16 / bfloat16, rife/__init__.py:120-134,195-198,210,
227-230,237-238) produce when the reference runs HERE (torch 2.10 CPU)?  This pins the decision recorded in DESIGN.md: the HIP
path computes in float32 for every ``dtype`` and reproduces the widget's I/O contract (frames rounded through the dtype,
float32 returned).

Findings (oracle/VALIDATION_DTYPE.log):
  * pass-through frames of the reference = input rounded through the dtype, returned as float32  -> reproduced bit-exactly;
  * every INTERPOLATED frame of the reference's float16 / bfloat16 CPU run is NaN: ``warp`` (rife_arch.py:31-70) feeds
    ``grid_sample`` a half-precision grid and this torch's CPU kernel returns NaN for it — so the reference offers no
    reduced-precision CPU execution to validate a reduced-precision device path against (north_star's oracle is the CPU path);
  * the reference's float32 result rounded once through the dtype is what the HIP node returns for that dtype
    (tests/test_gpu_rife.py::test_node_dtype_widget)."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import synth  # noqa: E402
from oracle import ref_import  # noqa: E402


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    sd = synth.rife47_synth_state_dict(1234)
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "rife47.pth")
        torch.save(sd, pth)
        R = ref_import.rife_node(pth)
        ok = True
        for (h, w, seed) in ((96, 128, 5), (256, 320, 7)):
            frames = synth.smooth_frames(3, h, w, seed=seed, shift=2.0)
            outs = {}
            for dt in ("float32", "float16", "bfloat16"):
                R._model_cache.clear()
                (outs[dt],) = R.RIFE_VFI().vfi("rife47.pth", frames, multiplier=2, dtype=dt)
            for dt in ("float16", "bfloat16"):
                o, td_ = outs[dt], getattr(torch, dt)
                src_ok = all(torch.equal(o[2 * i], frames[i].to(td_).float()) for i in range(3))
                nan_new = [torch.isnan(o[i]).float().mean().item() for i in (1, 3)]
                log(f"{h}x{w} dtype={dt}: returned dtype {o.dtype}; pass-through frames == input rounded through {dt}: {src_ok}; "
                    f"NaN fraction of the reference's interpolated frames: {nan_new}")
                ok &= src_ok and o.dtype == torch.float32
    log("DTYPE VALIDATION: I/O contract of the dtype widget pinned (pass-through rounding, float32 return); the reference's own "
        "half-precision CPU run yields NaN interpolated frames -> fp32 compute is the contract of the HIP path for every dtype"
        if ok else "DTYPE VALIDATION FAILED")
    with open(os.path.join(ROOT, "oracle", "VALIDATION_DTYPE.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
