"""Splat micro-benchmark of BASELINE.json configs[4] / SURVEY 8(d): in [1,4,1088,1920] U[0,1), flow N(0, 8 px) seed 2,
8 launches; algorithmic traffic 83.6 MB per launch = (4 in + 2 flow + 4 out) * 4 B * HW.  Also the M2M cost volumes."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from cfi_amd import _lib  # noqa: E402

MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # option splat_atomic (0 default; 3 = the staged list gather as one splat): needs the test build
if MODE:
    _lib.use_test_build()
ge.build()
lib = _lib.load()
if MODE:
    assert lib.vfi_test_set_option(b"splat_atomic", MODE) == 0
_lib.check(lib.vfi_init(0), "init")
p = lambda t: C.c_void_p(t.data_ptr())

if __name__ == "__main__":
    H, W, Cc = 1088, 1920, 4
    g = torch.Generator(device="cpu").manual_seed(2)
    for sigma in (8.0, 1.0, 0.1, 0.0, 32.0, -8.0, -24.0):
        x = torch.rand(1, H, W, Cc, generator=g).cuda()
        if sigma >= 0:      # i.i.d. N(0, sigma) per pixel — an incoherent field
            fl = (torch.randn(1, H, W, 2, generator=g) * sigma).cuda()
        else:               # a coherent field (what M2M's flow network produces): low-pass noise, amplitude |sigma| px
            base = torch.randn(1, 2, 9, 16, generator=g) * (-sigma)
            fl = torch.nn.functional.interpolate(base, size=(H, W), mode="bicubic", align_corners=True).permute(0, 2, 3, 1).contiguous().cuda()
        out = torch.empty_like(x)
        lib.vfi_softsplat_sum(p(x), p(fl), p(out), 1, H, W, Cc, None)
        torch.cuda.synchronize()
        lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
        for _ in range(8):
            lib.vfi_softsplat_sum(p(x), p(fl), p(out), 1, H, W, Cc, None)
        lib.vfi_trace_enable(0)
        rep = _lib.trace_report()
        calls = rep["softsplat_sum"][0]
        ms = sum(v[1] for v in rep.values())  # every pass of the launch
        bytes_ = (Cc + 2 + Cc) * 4 * H * W
        kind = f"iid sigma={sigma:4.1f}px" if sigma >= 0 else f"smooth amp={-sigma:4.1f}px"
        print(f"softsplat [1,{H},{W},{Cc}] flow {kind}: {ms / calls * 1e3:8.1f} us/launch  "
              f"{bytes_ / (ms / calls * 1e-3) / 1e9:8.1f} GB/s algorithmic ({bytes_ / 1e6:.1f} MB) "
              f"[all passes] " + " ".join(f"{k}={v[1] / v[0] * 1e3:.1f}us" for k, v in rep.items()), flush=True)
    for (h, w) in ((272, 480), (136, 240), (68, 120)):
        one = torch.randn(1, h, w, 32).cuda(); two = torch.randn(1, h, w, 32).cuda()
        out = torch.empty(1, h, w, 81).cuda()
        lib.vfi_costvol9x9(p(one), 32, p(two), 32, 0, p(out), 1, h, w, 32, 81, 0, None)
        torch.cuda.synchronize()
        lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
        for _ in range(8):
            lib.vfi_costvol9x9(p(one), 32, p(two), 32, 0, p(out), 1, h, w, 32, 81, 0, None)
        lib.vfi_trace_enable(0)
        calls, ms = _lib.trace_report()["costvol9x9"]
        bytes_ = (32 + 32 + 81) * 4 * h * w
        print(f"costvol9x9 [1,{h},{w},32]: {ms / calls * 1e3:8.1f} us/launch  {bytes_ / (ms / calls * 1e-3) / 1e9:8.1f} GB/s algorithmic", flush=True)
