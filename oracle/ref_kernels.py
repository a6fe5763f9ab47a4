"""ORACLE — test infrastructure only: run the PREBUILT host builds of the reference's own CUDA kernel text
(oracle/_ref/*.so, made in the build container by oracle/build_ref.py from /root/reference/vfi_models/ops/cupy_ops via the
reference's own ``cuda_kernel`` specialiser) without needing /root/reference — so the GPU box can compare the HIP
kernels with an execution of the reference kernels.  Shapes are fixed at build time (the reference specialises per shape);
oracle/_ref/manifest.json lists them.  Launch geometry and zero-initialisation restate the reference's launch sites:
softsplat.py:201-224 (out = zeros; grid = ceil(nelement/512), block = 512; args n, in, flow, out) and costvol.py:139-179
(out = empty [N,81,H,W]; n = N*H*W)."""
import ctypes as C
import json
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
_libs = {}


def manifest():
    p = os.path.join(_DIR, "manifest.json")
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        return json.load(f)


def available():
    m = manifest()
    return bool(m) and all(os.path.exists(os.path.join(_DIR, v)) for v in m.values())


def shapes(op):
    return [tuple(int(x) for x in k.split("|")[1].split(",")) for k in manifest() if k.startswith(op + "|")]


def _launch(op, shape, n, ptrs):
    so = manifest()[op + "|" + ",".join(str(s) for s in shape)]
    if so not in _libs:
        _libs[so] = C.CDLL(os.path.join(_DIR, so))
    fn = getattr(_libs[so], op + "__host")
    fn.restype = None
    keep = [C.c_int(n)] + [C.c_void_p(p) for p in ptrs]
    arr = (C.c_void_p * len(keep))(*[C.cast(C.pointer(k), C.c_void_p) for k in keep])
    fn(C.c_int((n + 511) // 512), C.c_int(1), C.c_int(1), C.c_int(512), C.c_int(1), C.c_int(1), arr)


def softsplat_out(ten_in, ten_flow):
    a = np.ascontiguousarray(ten_in, np.float32)
    f = np.ascontiguousarray(ten_flow, np.float32)
    out = np.zeros_like(a)
    _launch("softsplat_out", a.shape, out.size, [a.ctypes.data, f.ctypes.data, out.ctypes.data])
    return out


def costvol_out(one, two):
    a = np.ascontiguousarray(one, np.float32)
    b = np.ascontiguousarray(two, np.float32)
    n, c, h, w = a.shape
    out = np.empty((n, 81, h, w), np.float32)
    _launch("costvol_out", a.shape, n * h * w, [a.ctypes.data, b.ctypes.data, out.ctypes.data])
    return out
