"""GPU micro-benchmark: direct implicit-GEMM kernel vs the Winograd F(2x2,3x3) kernel (csrc/conv_wino.hip) on the 3x3 stride-1
layer shapes of the RIFE 4.7 trunk at 1080p (batch 32) and of FILM at 1080p; kernel time from the library's HIP-event tracing.
"effective" TFLOP/s = direct-form FLOP / time (what the layer is worth), "executed" = the MFMA FLOP the Winograd form issues
(direct / 2.25, without the padding of partial regions)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
ge.load_package()
from cfi_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.check(lib.vfi_init(0), "init")
PEAK = 157.3


def run(n, h, w, cin, cout, variant, reps=4):
    g = torch.Generator().manual_seed(cin * 7 + cout + h)
    x = (torch.rand(n, h, w, cin, generator=g) - 0.5).cuda()
    wt = (torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * 0.1
    b = torch.rand(cout, generator=g) - 0.5
    out = torch.empty(n, h, w, cout, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if lib.vfi_conv3x3(p(x), p(wt), p(b), None, p(out), n, h, w, cin, cout, 1, 1, 0.2, variant, None):
        return None, None
    lib.vfi_trace_reset()
    lib.vfi_trace_enable(1)
    for _ in range(reps):
        lib.vfi_conv3x3(p(x), p(wt), p(b), None, p(out), n, h, w, cin, cout, 1, 1, 0.2, variant, None)
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    calls, tot = list(rep.values())[0]
    return tot / calls, out


LAYERS = [("rife res_c64 x32", 32, 272, 480, 64, 64), ("rife res_c96 x32", 32, 136, 240, 96, 96), ("rife res_c128 x32", 32, 68, 120, 128, 128),
          ("rife res_c192 x32", 32, 34, 60, 192, 192), ("rife res_c64 x8", 8, 272, 480, 64, 64),
          ("film 64->64 @1080p", 1, 1080, 1920, 64, 64), ("film 200->64 @1080p", 1, 1080, 1920, 200, 64), ("film 128->128 @540p", 2, 540, 960, 128, 128),
          ("film 520->128 @540p", 1, 540, 960, 520, 128), ("film 256->256 @270p", 2, 270, 480, 256, 256), ("film 512->512 @135p", 1, 135, 240, 512, 512),
          ("film 2440->512 @135p", 1, 135, 240, 2440, 512), ("m2m 128->128 @272x480", 2, 272, 480, 128, 128)]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if any(k in l[0] for k in sys.argv[1:])]
print(f"{'layer':28s} {'direct ms':>10s} {'TF/s':>7s} | {'wino16x8 ms':>11s} {'eff TF/s':>8s} {'exec frac':>9s} | {'wino32x4 ms':>11s} | max|d| vs direct")
for name, n, h, w, cin, cout in LAYERS:
    flop = 2.0 * n * h * w * cin * cout * 9
    t0, o0 = run(n, h, w, cin, cout, -1)
    t1, o1 = run(n, h, w, cin, cout, 100)
    t2, o2 = run(n, h, w, cin, cout, 101)
    d = lambda o: (o - o0).abs().max().item() if o is not None else -1
    f = lambda t: f"{flop / t / 1e9:7.1f}" if t else "   n/a"
    ex = lambda t: flop / 2.25 / t / 1e9 / PEAK if t else 0
    print(f"{name:28s} {t0:10.4f} {f(t0)} | {t1 or 0:11.4f} {f(t1):>8s} {ex(t1):9.3f} | {t2 or 0:11.4f} | {d(o1):.2e} {d(o2):.2e}", flush=True)
