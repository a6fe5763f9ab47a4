"""ORACLE — test infrastructure only: ctypes wrapper of oracle/m2m_ops.c (softsplat sum, 9x9 cost volume).

Parity status: PINNED BY EXECUTION of the reference's kernel text (oracle/validate_m2m_vs_reference.py: the reference's own
cupy_ops package on a host shim, bit-exact; oracle/VALIDATION_M2M.log, tests/golden/m2m_ops_ref.npz, oracle/_ref).  The C
code restates vfi_models/ops/cupy_ops/softsplat.py:140-192 and costvol.py:4-43 line by line."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libvfi_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "m2m_ops.c")):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(so)
        for f in (_lib.oracle_softsplat_sum, _lib.oracle_softsplat_sum_rev, _lib.oracle_costvol):
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    return _lib


def softsplat_sum(ten_in, ten_flow, reverse=False):
    """NCHW float32 numpy arrays -> summation splat (softsplat_func.forward, cupy_ops/softsplat.py:197-233).  ``reverse``: the same
    contributions added in the opposite source order (oracle/m2m_hot_certificate.py)."""
    a = np.ascontiguousarray(ten_in, np.float32)
    f = np.ascontiguousarray(ten_flow, np.float32)
    n, c, h, w = a.shape
    assert f.shape == (n, 2, h, w)
    out = np.empty_like(a)
    (lib().oracle_softsplat_sum_rev if reverse else lib().oracle_softsplat_sum)(a.ctypes.data, f.ctypes.data, out.ctypes.data, n, c, h, w)
    return out


def costvol(one, two):
    """NCHW float32 -> [N,81,H,W] (costvol_func.forward, cupy_ops/costvol.py:135-183)."""
    a = np.ascontiguousarray(one, np.float32)
    b = np.ascontiguousarray(two, np.float32)
    n, c, h, w = a.shape
    assert b.shape == a.shape
    out = np.empty((n, 81, h, w), np.float32)
    lib().oracle_costvol(a.ctypes.data, b.ctypes.data, out.ctypes.data, n, c, h, w)
    return out
