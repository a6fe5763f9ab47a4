"""GPU micro-benchmark of the 1x1 layer objects on GMFSS's transformer shapes (1080p: 130560 / 16320 tokens) for a set of tile
variants (forced per trace name through vfi_test_variant_override, include/vfi_hip_test.h)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(4, 136, 240, 128, 128, 0), (4, 136, 240, 256, 1024, 5), (4, 136, 240, 1024, 128, 0), (2, 68, 120, 128, 128, 0),
          (2, 68, 120, 256, 1024, 5), (2, 68, 120, 1024, 128, 0), (1, 1080, 1920, 32, 16, 0), (1, 135, 240, 256, 128, 0)]


def child():
    import torch
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    from cfi_amd import _lib
    _lib.use_test_build()      # vfi_test_variant_override lives in libvfi_hip_test.so only
    lib = _lib.load()
    out = []
    for n, h, w, cin, cout, act in SHAPES:
        x = torch.randn(n, h, w, cin, device="cuda")
        wt, b = torch.randn(cout, cin, 1, 1) * 0.05, torch.randn(cout) * 0.1
        hnd = lib.vfi_conv_create_ex(0, wt.data_ptr(), b.data_ptr(), cout, cin, 1, 1, 0, None, cin, None)
        o = torch.empty(n, h, w, cout, device="cuda")
        call = lambda: _lib.check(lib.vfi_conv_forward_ex(hnd, x.data_ptr(), cin, h, w, o.data_ptr(), cout, n, act, 0.0, 0.0, 0.0, None, 0, None), "fwd")
        try:
            call()
        except RuntimeError:
            out.append(f"{cin}->{cout}@{n}x{h}x{w}: n/a")
            lib.vfi_conv_destroy(hnd)
            continue
        torch.cuda.synchronize()
        lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        lib.vfi_trace_enable(0)
        rep = _lib.trace_report()
        ms = sum(v[1] for v in rep.values()) / 5
        out.append(f"{cin}->{cout}@{n}x{h}x{w}: {ms * 1e3:7.1f} us {2 * n * h * w * cin * cout / ms / 1e9:6.1f} TF")
        lib.vfi_conv_destroy(hnd)
    print(" | ".join(out), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.load_package()
    from cfi_amd import _lib as _L
    names = [f"conv1x1s1_{cin}to{cout}" for _, _, _, cin, cout, _ in SHAPES]
    for label, var in [("picker", None), ("m2n2 k8 (48)", 48), ("m1n2 k8 (49)", 49), ("m1n2 k32 (55)", 55), ("m2n2w22 k32 (56)", 56)]:
        spec = ",".join(f"{n}={var}" for n in sorted(set(names))) if var is not None else ""
        _L.load().vfi_test_variant_override(spec.encode())        # A/B hook of include/vfi_hip_test.h (tile variant by trace name)
        print(f"{label:16s} ", end="", flush=True)
        child()
