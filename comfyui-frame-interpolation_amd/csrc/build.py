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
LIB = os.path.join(PKG, "libvfi_hip.so")                # the product library: exactly the entry points of include/vfi_hip.h
LIB_TEST = os.path.join(PKG, "libvfi_hip_test.so")      # the same objects + the test taps of include/vfi_hip_test.h (-DVFI_TEST_TAPS): tests/ and tools/ only
STAMP = LIB + ".stamp"
TAPS_MACRO = "VFI_TEST_TAPS"
ARCH = "gfx950"
# per-file flags.  conv_wino.hip: hipcc's SLP vectoriser packs the input-transform adds into v_pk_add_f32 with v_mov shuffles
# — packed f32 VALU beside MFMAs is an anti-lever on gfx950 (MI355X_MICROARCH.md, per-instruction constants) and costs registers
EXTRA_FLAGS = {"conv_wino.hip": ["-fno-slp-vectorize"],
               # cost volume: SLP turned |a - b| accumulation (v_sub + v_add with the |x| source modifier, 2 ops) into v_pk_add + 2 v_and +
               # v_pk_add + moves (~3 per element)
               "m2m_ops.hip": ["-fno-slp-vectorize", "-ffp-contract=off"],
               # M2M's element-wise kernels restate torch's unfused fp32 sequences with __fmul_rn / __fadd_rn — which HIP defines as plain
               # `x * y` / `x + y`, so the default -ffp-contract=fast fused them anyway (a grid coordinate lin + flow * scale as one fma moves
               # a warp tap by an ulp of 1000 px = 6e-5 px).  Off for these files: the sequences now are what their comments say.
               "m2m_net.hip": ["-ffp-contract=off"],
               # the one-kernel M2M render sums in * w products exactly as the reference's atomicAdd(out, in * w): no mul + add may become an
               # fma (a function-scope `#pragma clang fp contract(off)` did not survive inlining: 94 v_fmac_f32 in the first build)
               "m2m_render.hip": ["-fno-slp-vectorize", "-ffp-contract=off"]}


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


def _has_taps(src):
    with open(src) as fh:
        return TAPS_MACRO in fh.read()


def build_lib(force=False, verbose=True):
    """Builds BOTH libraries; returns the product one.  Translation units that carry test taps (`#ifdef VFI_TEST_TAPS`) are compiled
    twice — without the macro for libvfi_hip.so, with it for libvfi_hip_test.so; every other object is shared.  The product library
    therefore has no entry point that changes a kernel choice (vfi_test_set_option, vfi_test_variant_override, vfi_test_conv_algo) and
    none of the read-back taps."""
    d = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(LIB_TEST) and os.path.exists(STAMP) and open(STAMP).read().strip() == d:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, objs_test = [], []
    procs = []
    for src in _sources():
        base = os.path.basename(src)
        obj = os.path.join(CSRC, base[:-4] + ".o")
        common = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(base, [])
        jobs = [(common + ["-c", src, "-o", obj], obj)]
        objs.append(obj)
        if _has_taps(src):
            tobj = os.path.join(CSRC, base[:-4] + ".taps.o")
            jobs.append((common + ["-D" + TAPS_MACRO, "-c", src, "-o", tobj], tobj))
            objs_test.append(tobj)
        else:
            objs_test.append(obj)
        for cmd, _ in jobs:
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((subprocess.Popen(cmd), src))
    for p, src in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    for lib, ob in ((LIB, objs), (LIB_TEST, objs_test)):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + ob + ["-ldl"]   # dl: comm.hip binds RCCL at first use
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(d + "\n")
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
