"""ORACLE tooling — a host stand-in for ``cupy`` so that the reference's OWN CuPy op package
(/root/reference/vfi_models/ops/cupy_ops) imports and RUNS in this CPU-only container.

The reference builds each kernel by specialising a CUDA source string in Python (cupy_ops/utils.py:29-213:
``{{type}}`` / ``SIZE_n`` / ``OFFSET_n`` / ``VALUE_n`` are replaced by literals for the given tensors) and hands the text
to ``cupy.RawModule`` (utils.py:242).  Here ``RawModule`` compiles exactly that specialised text with g++ behind a
serial execution shim (``__global__`` -> plain function, ``blockIdx`` / ``threadIdx`` -> loop variables, ``atomicAdd`` ->
``+=``): one loop iteration is one CUDA thread, visited in global-thread-index order.  On a GPU the atomics commit in
an unspecified order; the serial order is one of the admissible ones, so outputs are *an* execution of the reference
kernel, deterministic and bit-reproducible.

Only kernels without ``__shared__`` / ``__syncthreads`` can be serialised thread by thread (softsplat_out, costvol_out
and their gradients qualify); anything else raises.

Generated sources and shared objects go to oracle/_ref/ (git-ignored: the generated text contains reference code and
must stay out of history; the .so files travel to the GPU box with the snapshot).  Nothing under the product package,
bench.py's timed path or smoke()'s device path imports this.
"""
import ctypes
import hashlib
import os
import re
import subprocess

import numpy as np

int32 = np.int32
float32 = np.float32

_REF_DIR = os.environ.get("VFI_ORACLE_REF_DIR") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "_ref")

_SHIM = r"""
// serial CUDA-on-host shim (oracle tooling, see oracle/stubs/cupy/__init__.py)
#include <cassert>
#include <cmath>
#include <cstdlib>
#include <cstdint>
using namespace std;   // CUDA's overloaded abs / floor / isfinite on float == the std:: float overloads
#define __global__
#define __device__
#define __launch_bounds__(...)
struct vfi_dim3 { int x, y, z; };
static vfi_dim3 blockIdx, blockDim, threadIdx, gridDim;
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
"""


built = []   # (kernel names, .so path) of every module instantiated in this process, in order (oracle/build_ref.py)


def memoize(for_each_device=False):
    def deco(fn):
        cache = {}

        def wrapper(*a):
            if a not in cache:
                cache[a] = fn(*a)
            return cache[a]

        return wrapper

    return deco


_SIG = re.compile(r'extern\s+"C"\s+__global__\s+void\s+(?:__launch_bounds__\([^)]*\)\s*)?(\w+)\s*\(([^)]*)\)', re.S)


def _driver(code):
    """The host launcher for every kernel in ``code``: loops blocks and threads in index order."""
    out = []
    for name, params in _SIG.findall(code):
        types = []
        for p in params.split(","):
            p = p.strip().replace("__restrict__", "")
            types.append(p[: re.search(r"\w+\s*$", p).start()].strip())
        args = ", ".join(f"*({t}*)args[{i}]" for i, t in enumerate(types))
        out.append(f"""
extern "C" void {name}__host(int gx, int gy, int gz, int bx, int by, int bz, void** args) {{
    gridDim = vfi_dim3{{gx, gy, gz}}; blockDim = vfi_dim3{{bx, by, bz}};
    for (int Bz = 0; Bz < gz; ++Bz) for (int By = 0; By < gy; ++By) for (int Bx = 0; Bx < gx; ++Bx)
    for (int Tz = 0; Tz < bz; ++Tz) for (int Ty = 0; Ty < by; ++Ty) for (int Tx = 0; Tx < bx; ++Tx) {{
        blockIdx = vfi_dim3{{Bx, By, Bz}}; threadIdx = vfi_dim3{{Tx, Ty, Tz}};
        {name}({args});
    }}
}}""")
    return "\n".join(out)


class _HostKernel:
    def __init__(self, lib, name):
        self.fn = getattr(lib, name + "__host")
        self.fn.restype = None

    def __call__(self, grid, block, args, stream=None, shared_mem=0):
        assert shared_mem == 0
        g, b = (tuple(grid) + (1, 1, 1))[:3], (tuple(block) + (1, 1, 1))[:3]
        keep = []
        for a in args:
            if isinstance(a, np.int32):
                keep.append(ctypes.c_int(int(a)))
            elif isinstance(a, np.float32):
                keep.append(ctypes.c_float(float(a)))
            elif isinstance(a, int):                       # tensor.data_ptr()
                keep.append(ctypes.c_void_p(a))
            else:
                raise TypeError(f"kernel argument of type {type(a)}")
        arr = (ctypes.c_void_p * len(keep))(*[ctypes.cast(ctypes.pointer(k), ctypes.c_void_p) for k in keep])
        self.fn(*[ctypes.c_int(int(v)) for v in g + b], arr)


class RawModule:
    """cupy.RawModule(code=...) — g++ instead of NVRTC; the result is cached by the hash of the specialised text."""

    def __init__(self, code, **kw):
        if "__shared__" in code or "__syncthreads" in code:
            raise NotImplementedError("host shim: kernels with shared memory / barriers cannot be serialised per thread")
        os.makedirs(_REF_DIR, exist_ok=True)
        names = "_".join(n for n, _ in _SIG.findall(code))
        tag = f"{names}_{hashlib.sha1(code.encode()).hexdigest()[:16]}"
        so = os.path.join(_REF_DIR, tag + ".so")
        if not os.path.exists(so):
            src = os.path.join(_REF_DIR, tag + ".cpp")
            with open(src, "w") as f:
                f.write(_SHIM + code + _driver(code))
            tmp = so + f".{os.getpid()}.tmp"
            subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17", "-w", "-o", tmp, src])
            os.replace(tmp, so)
        self.path = so
        built.append((names, so))
        self.lib = ctypes.CDLL(so)

    def get_function(self, name):
        return _HostKernel(self.lib, name)
