"""GPU tool: FILM forward at 1080p with direct-conv tile variants forced per layer shape (trace names carry the shape under
VFI_TRACE_SHAPES=1; include/vfi_hip_test.h: vfi_test_variant_override) — HIP-event ms of the layers named in the overrides.
    VFI_TRACE_SHAPES=1 python tools/film_variant_ab.py "conv3x3_1920to256@67x120=36" ..."""
import os
import sys
import time

import torch

os.environ.setdefault("VFI_TRACE_SHAPES", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from cfi_amd import _lib, synth  # noqa: E402

_lib.use_test_build()      # the override lives in libvfi_hip_test.so only
ge.build()
lib = _lib.load()
from cfi_amd.film import FilmEngine  # noqa: E402

H, W = 1080, 1920
fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
eng = FilmEngine(synth.film_synth_state_dict(1234))
watch = sorted({kv.split("=")[0] for spec in sys.argv[1:] for kv in spec.split(",") if "=" in kv})
ref = None
for spec in [""] + sys.argv[1:]:
    lib.vfi_test_variant_override(spec.encode())
    try:
        eng.forward(x0, x1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = eng.forward(x0, x1)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 3 * 1e3
        lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
        out = eng.forward(x0, x1)
        torch.cuda.synchronize()
        lib.vfi_trace_enable(0)
        rep = _lib.trace_report()
        lib.vfi_trace_reset()
        o = out[::8, ::8].clone()
        ref = o if ref is None else ref
        print(f"[{spec or 'default'}] forward {wall:.2f} ms  " + "  ".join(f"{k}={rep[k][1]:.3f}/{rep[k][0]}" for k in watch if k in rep) +
              f"  max|d| vs default {float((o - ref).abs().max()):.1e}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"[{spec}] FAILED: {e}", flush=True)
lib.vfi_test_variant_override(b"")
