"""GPU micro-benchmark: every compatible tile variant of csrc/conv_mfma.hip on the RIFE 4.7 trunk
layer shapes (1080p), kernel time from the library's HIP-event tracing.  Prints a table and the
fp32-MFMA roofline fraction; used to set conv_pick_variant()."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from cfi_amd import _lib as _vfi_lib  # noqa: E402

_vfi_lib.use_test_build()      # the A/B taps live in libvfi_hip_test.so only; one process uses one library, chosen before build() loads it
ge.build()
ge.load_package()
from cfi_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.check(lib.vfi_init(0), "init")

VARIANTS = {  # index: (stride, BN, CK)
    0: (1, 64, 16), 1: (1, 96, 16), 2: (1, 64, 16), 3: (1, 96, 16), 4: (1, 32, 16), 5: (1, 32, 16), 6: (1, 64, 16),
    7: (1, 128, 16), 8: (2, 64, 8), 9: (2, 96, 8), 10: (2, 32, 8), 11: (2, 64, 8),
    32: (1, 64, 8), 33: (1, 96, 8), 34: (1, 64, 8), 35: (1, 96, 8), 36: (1, 128, 8), 37: (1, 128, 8), 38: (1, 64, 16),
    39: (2, 64, 8), 40: (2, 64, 8), 41: (2, 96, 8), 42: (2, 32, 8),
}
NAMES = {0: "s1_m2n2", 1: "s1_m2n3", 2: "s1_m1n2", 3: "s1_m1n3", 4: "s1_m1n1", 5: "s1_m2n1", 6: "s1_m1n1w22",
         7: "s1_m2n2w22", 8: "s2_m1n2", 9: "s2_m1n3", 10: "s2_m1n1", 11: "s2_m2n2", 32: "d1_m2n2", 33: "d1_m2n3",
         34: "d1_m1n2", 35: "d1_m1n3", 36: "d1_m2n2w22", 37: "d1_m4n2w22", 38: "d1_m2n2k16", 39: "d2_m1n2",
         40: "d2_m2n2", 41: "d2_m1n3", 42: "d2_m2n1"}


def run(n, h, w, cin, cout, stride, res, variant, reps=5):
    x = torch.rand(n, h, w, cin, device="cuda") - 0.5
    wt = (torch.rand(cout, cin, 3, 3) - 0.5) * 0.1
    b = torch.rand(cout) - 0.5
    beta = torch.rand(cout) + 0.5 if res else None
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    out = torch.empty(n, ho, wo, cout, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    if lib.vfi_conv3x3(p(x), p(wt), p(b), p(beta), p(out), n, h, w, cin, cout, stride, 1, 0.2, variant, None):
        return None  # (also the untimed warm-up call)
    lib.vfi_trace_reset()
    lib.vfi_trace_enable(1)
    for _ in range(reps):
        rc = lib.vfi_conv3x3(p(x), p(wt), p(b), p(beta), p(out), n, h, w, cin, cout, stride, 1, 0.2, variant, None)
        if rc:
            lib.vfi_trace_enable(0)
            return None
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    ms = sorted(v[1] / v[0] for v in rep.values())[0] if rep else None
    # per-call times are aggregated; use the mean of the last reps-1 by re-measuring total
    calls, tot = list(rep.values())[0]
    return tot / calls


LAYERS = [  # name, H, W, cin, cout, stride, res   (trunk resolution at 1080p = padded 1088x1920)
    # res=False: in the network the ResConv residual is folded into the centre tap (rife_net.hip)
    ("res_c64", 272, 480, 64, 64, 1, False),
    ("res_c96", 136, 240, 96, 96, 1, False),
    ("res_c128", 68, 120, 128, 128, 1, False),
    ("res_c192", 34, 60, 192, 192, 1, False),
    ("c0a_b3", 1088, 1920, 24, 32, 2, False),
    ("c0b_b3", 544, 960, 32, 64, 2, False),
    ("c0a_b2", 544, 960, 24, 48, 2, False),
    ("c0b_b2", 272, 480, 48, 96, 2, False),
    ("c0a_b1", 272, 480, 24, 64, 2, False),
    ("c0b_b1", 136, 240, 64, 128, 2, False),
    ("c0a_b0", 136, 240, 16, 96, 2, False),
    ("c0b_b0", 68, 120, 96, 192, 2, False),
]

def run_deconv(n, h, w, cin, reps=5):
    x = torch.rand(n, h, w, cin, device="cuda") - 0.5
    wt = (torch.rand(cin, 24, 4, 4) - 0.5) * 0.1
    b = torch.rand(24) - 0.5
    out = torch.empty(n, 4 * h, 4 * w, 6, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    lib.vfi_deconv4x4_ps2(p(x), p(wt), p(b), p(out), n, h, w, cin, 24, None)  # untimed warm-up
    lib.vfi_trace_reset()
    lib.vfi_trace_enable(1)
    for _ in range(reps):
        if lib.vfi_deconv4x4_ps2(p(x), p(wt), p(b), p(out), n, h, w, cin, 24, None):
            lib.vfi_trace_enable(0)
            return None
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    calls, tot = list(rep.values())[0]
    return tot / calls


if __name__ == "__main__":
    batches = [int(a) for a in sys.argv[1:]] or [1, 8]
    from cfi_amd import _lib as _L
    for B in batches:
        for gv in (12, 13, 43, 44):      # grouped (transposed-conv) tile variants, forced through the A/B option (include/vfi_hip_test.h)
            assert _L.load().vfi_test_set_option(b"grouped_variant", gv) == 0
            for name, h, w, cin in (("last_b3", 272, 480, 64), ("last_b2", 136, 240, 96), ("last_b1", 68, 120, 128), ("last_b0", 34, 60, 192)):
                ms = run_deconv(B, h, w, cin)
                flop = 2.0 * B * h * w * cin * 24 * 16
                print(f"{name:10s} {B:2d} grouped_v{gv:<3d} {ms:8.4f} {flop / (ms * 1e-3) / 1e12:8.2f} (algorithmic TFLOP/s)", flush=True)
        _L.load().vfi_test_set_option(b"grouped_variant", -1)
    print(f"{'layer':10s} {'B':>2s} {'variant':12s} {'ms':>8s} {'TFLOP/s':>8s} {'frac':>6s}")
    for name, h, w, cin, cout, stride, res in LAYERS:
        for B in batches:
            ho, wo = h // stride, w // stride
            flop = 2.0 * B * ho * wo * cin * cout * 9
            best = None
            for v, (vs, bn, ck) in VARIANTS.items():
                if vs != stride or (-(-cout // 32) * 32) % bn or cin % ck:
                    continue
                ms = run(B, h, w, cin, cout, stride, res, v)
                if ms is None:
                    continue
                tf = flop / (ms * 1e-3) / 1e12
                print(f"{name:10s} {B:2d} {NAMES[v]:12s} {ms:8.4f} {tf:8.2f} {tf / 157.3:6.3f}", flush=True)
                if best is None or ms < best[1]:
                    best = (v, ms)
            if best:
                print(f"  -> best for {name} B={B}: {NAMES[best[0]]} ({best[1]:.4f} ms)", flush=True)
