"""ORACLE tooling (build container only) — prebuild host executables of the reference's own CUDA kernels for a fixed list
of shapes into oracle/_ref/ (+ manifest.json), so that tests on the GPU box — where /root/reference does not exist — can
still check the HIP kernels and oracle/m2m_ops.c against an EXECUTION of the reference kernel text.

Recipe: import the reference's ``vfi_models.ops`` (config.yaml selects cupy_ops) with oracle/stubs/cupy standing in for
CuPy; call the reference's own ``softsplat_func.apply`` / ``costvol_func.apply`` once per shape: the reference's
``cuda_kernel`` (cupy_ops/utils.py:29-213) specialises the kernel string, the stub's ``RawModule`` compiles that text with
g++ behind a serial ``__global__`` / ``atomicAdd`` shim.  Outputs go to oracle/_ref/ only (git-ignored; travels with gpurun).
Run by ``__graft_entry__.build()`` when /root/reference is present (as a subprocess: it patches torch.Tensor.is_cuda)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# NCHW shapes of tenIn (softsplat) / tenOne (costvol) the tests use
# (r5: + SURVEY 8d config 5's splat stress shape [1,4,1088,1920] and the two finest cost-volume levels of M2M at 1080p, so that every
# level 17x30 ... 272x480 and the benchmarked splat size are checked against an execution of the reference's kernel text)
SOFTSPLAT_SHAPES = [(1, 4, 64, 96), (2, 4, 50, 70), (2, 3, 33, 47), (8, 4, 128, 192), (1, 4, 272, 480), (1, 4, 1088, 1920)]
COSTVOL_SHAPES = [(2, 32, 17, 30), (2, 32, 34, 60), (1, 32, 20, 24), (2, 32, 68, 120), (2, 32, 136, 240), (2, 32, 272, 480)]


def main():
    import warnings

    import torch

    warnings.simplefilter("ignore")
    from oracle import ref_import

    if not ref_import.available():
        print("oracle/build_ref.py: /root/reference not present - nothing to do")
        return 0
    ops = ref_import.reference_ops()
    import cupy   # oracle/stubs/cupy

    man = {}
    for shp in SOFTSPLAT_SHAPES:
        n, c, h, w = shp
        ops.softsplat_func.apply(torch.zeros(shp), torch.zeros(n, 2, h, w))
        names, so = cupy.built[-1]
        assert names == "softsplat_out"
        man["softsplat_out|" + ",".join(map(str, shp))] = os.path.basename(so)
    for shp in COSTVOL_SHAPES:
        ops.costvol_func.apply(torch.zeros(shp), torch.zeros(shp))
        names, so = cupy.built[-1]
        assert names == "costvol_out"
        man["costvol_out|" + ",".join(map(str, shp))] = os.path.basename(so)
    ref_dir = os.path.dirname(cupy.built[-1][1])
    with open(os.path.join(ref_dir, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1)
    print(f"oracle/_ref: {len(man)} reference kernels built for the host")
    return 0


if __name__ == "__main__":
    sys.exit(main())
