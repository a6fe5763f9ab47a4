"""GPU tool: A/B of the transposed convolutions' two forms (grouped direct kernel vs ONE 3x3 layer on the Winograd kernel, test option
deconv_wino) inside M2M, IFUNet, IFRNet_L and RIFE 4.0 at 1080p — same process, same box, alternating, per-shape trace rows."""
import os
import sys
import time

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

lib = _lib.load()
H, W = 1080, 1920
fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
out = torch.empty(H, W, 3, device="cuda")


def timed(fn, n=6):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def deconv_rows(fn):
    lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
    fn()
    torch.cuda.synchronize()
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    lib.vfi_trace_reset()
    return {k: v[1] / v[0] for k, v in rep.items() if k.startswith("deconv")}, sum(v[1] for v in rep.values())


def ab(name, fn):
    res = {0: [], 1: []}
    for rep in range(3):
        for opt in (1, 0):
            lib.vfi_test_set_option(b"deconv_wino", opt)
            res[opt].append(timed(fn))
    rows = {}
    for opt in (1, 0):
        lib.vfi_test_set_option(b"deconv_wino", opt)
        rows[opt] = deconv_rows(fn)
    lib.vfi_test_set_option(b"deconv_wino", 1)
    print(f"{name}: Winograd form {min(res[1]):.2f} ms (runs {', '.join(f'{t:.2f}' for t in res[1])}) | direct form {min(res[0]):.2f} ms ({', '.join(f'{t:.2f}' for t in res[0])}); "
          f"traced sums {rows[1][1]:.2f} / {rows[0][1]:.2f} ms", flush=True)
    for k in sorted(rows[1][0]):
        print(f"     {k:36s} winograd {rows[1][0][k] * 1e3:8.1f} us   direct {rows[0][0].get(k, float('nan')) * 1e3:8.1f} us", flush=True)


from cfi_amd.m2m import M2MEngine  # noqa: E402

e = M2MEngine(synth.m2m_synth_state_dict(1234))
ab("M2M prepare", lambda: e.prepare(x0, x1))
e.close()
from cfi_amd.ifunet import IFUNetEngine  # noqa: E402

e = IFUNetEngine(synth.ifunet_synth_state_dict(1234))
ab("IFUNet forward (ensemble)", lambda: e.forward(x0, x1, 0.5, out, scale=1.0, ensemble=True))
e.close()
from cfi_amd.ifrnet import IFRNetEngine  # noqa: E402

e = IFRNetEngine(synth.ifrnet_synth_state_dict("L", 1234), "L")
o4 = out.view(1, H, W, 3)
ab("IFRNet_L node default", lambda: e.forward([x0], [x1], 0.5, 1.0, o4))
ab("IFRNet_L full resolution", lambda: e.forward([x0], [x1], 1.0, 0.5, o4))
e.close()
