"""TEST INFRASTRUCTURE — host-side build of the GMFSS / IFUNet kernel bodies.

``csrc/gmfss_ops.hip`` and ``csrc/ifunet_ops.hip`` compiled with ``-DVFI_HOSTCHECK`` exports the same C-ABI entry points as libvfi_hip.so, but each one
runs its per-element body (``csrc/gmfss_bodies.h``, ``__host__ __device__``) in a plain host loop instead of launching the
kernel.  No GPU is needed to build or run it, pointers are host pointers.  The CPU test suite uses it (a) to check every
body against torch and (b) as the backend of the test double of the C ABI (tests/emu_backend.py) that runs the engine's
orchestration on the CPU.  Nothing in the package imports this module.
"""
import ctypes as C
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "comfyui-frame-interpolation_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libvfi_hostcheck.so")
SOURCES = [os.path.join(CSRC, "gmfss_ops.hip"), os.path.join(CSRC, "ifunet_ops.hip"), os.path.join(CSRC, "util.hip")]
DEPS = SOURCES + [os.path.join(CSRC, h) for h in ("gmfss_bodies.h", "ifunet_bodies.h", "body_launch.h", "vfi_common.h")] + [os.path.join(ROOT, "include", "vfi_hip.h")]
_lib = None


def _digest():
    h = hashlib.sha256()
    for f in DEPS:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build():
    os.makedirs(OUT, exist_ok=True)
    stamp = LIB + ".stamp"
    d = _digest()
    if os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == d:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-DVFI_HOSTCHECK", "-o", LIB] + SOURCES)
    with open(stamp, "w") as f:
        f.write(d + "\n")
    return LIB


def load():
    """ctypes handle with the prototypes of cfi_amd._lib applied to every symbol the library exports."""
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (its bundled HIP runtime must be the one in the process, as for libvfi_hip.so)
        from cfi_amd import _lib as L

        lib = C.CDLL(build())
        for name, (res, args) in L.PROTOTYPES.items():
            if hasattr(lib, name):
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib
