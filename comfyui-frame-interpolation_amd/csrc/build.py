"""Ahead-of-time build of libvfi_hip.so for gfx950 (MI355X).  No JIT, no torch extension:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC csrc/*.hip -o libvfi_hip.so

hipcc cross-compiles without a GPU present; the .so is kept in-tree (git-ignored) so it travels
with the repository snapshot to the GPU box.
"""
import glob
import hashlib
import os
import subprocess
import sys

CSRC = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(CSRC)
LIB = os.path.join(PKG, "libvfi_hip.so")
STAMP = LIB + ".stamp"
ARCH = "gfx950"
# per-file flags.  conv_wino.hip: hipcc's SLP vectoriser packs the input-transform adds into v_pk_add_f32 with v_mov shuffles
# — packed f32 VALU beside MFMAs is an anti-lever on gfx950 (MI355X_MICROARCH.md, per-instruction constants) and costs registers
EXTRA_FLAGS = {"conv_wino.hip": ["-fno-slp-vectorize"],
               # cost volume: SLP turned |a - b| accumulation (v_sub + v_add with the |x| source modifier, 2 ops) into v_pk_add + 2 v_and +
               # v_pk_add + moves (~3 per element)
               "m2m_ops.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    files = _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.abspath(__file__)]
    files.append(os.path.join(PKG, "..", "include", "vfi_hip.h"))
    files.append(os.path.join(PKG, "..", "include", "vfi_hip_test.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode())  # names only: the checkout path differs on the GPU box
            h.update(fh.read())
    return h.hexdigest()


def build_lib(force=False, verbose=True):
    d = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == d:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in _sources():
        obj = os.path.join(CSRC, os.path.basename(src)[:-4] + ".o")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), src))
        objs.append(obj)
    for p, src in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]   # dl: comm.hip binds RCCL at first use
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(d + "\n")
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
