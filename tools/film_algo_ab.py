"""GPU tool: per-shape A/B of FILM's convolution layers at 1080p — automatic algorithm choice (Winograd where eligible) vs the direct
implicit-GEMM kernel everywhere (test hook vfi_test_conv_algo) — same process, same box.  Prints the rows whose two forms differ by > 5 %."""
import os
import sys

os.environ["VFI_TRACE_SHAPES"] = "1"
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from cfi_amd import _lib as _vfi_lib  # noqa: E402

_vfi_lib.use_test_build()      # the A/B taps live in libvfi_hip_test.so only; one process uses one library, chosen before build() loads it
ge.build()
ge.load_package()
from cfi_amd import _lib, synth  # noqa: E402
from cfi_amd.film import FilmEngine  # noqa: E402

lib = _lib.load()
H, W = 1080, 1920
eng = FilmEngine(synth.film_synth_state_dict(1234))
fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
rows = {}
for algo in (0, 1, 0, 1):
    lib.vfi_test_conv_algo(algo)
    eng.forward(x0, x1)
    torch.cuda.synchronize()
    lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
    eng.forward(x0, x1)
    torch.cuda.synchronize()
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    lib.vfi_trace_reset()
    for k, v in rep.items():
        rows.setdefault(k, {}).setdefault(algo, []).append(v[1])
    print(f"algo {algo}: traced total {sum(v[1] for v in rep.values()):.2f} ms", flush=True)
lib.vfi_test_conv_algo(0)
print(f"{'layer':34s} {'auto ms':>9s} {'direct ms':>10s}")
for k, d in sorted(rows.items(), key=lambda kv: -min(kv[1].get(0, [0]))):
    if 0 in d and 1 in d and k.startswith("conv"):
        a, b = min(d[0]), min(d[1])
        if abs(a - b) > 0.05 * max(a, b):
            print(f"{k:34s} {a:9.3f} {b:10.3f}   {'DIRECT better' if b < a else ''}")
