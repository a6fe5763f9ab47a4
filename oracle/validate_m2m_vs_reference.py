"""ORACLE tooling — pin oracle/m2m_ops.c and oracle/m2m_model_oracle.py against EXECUTIONS of the reference, here, on CPU.

The reference module imports vfi_models.ops (config.yaml: the CuPy backend).  oracle/stubs/cupy stands in for CuPy: the
reference's own ``cuda_kernel`` specialises its own kernel strings (cupy_ops/utils.py:29-213, softsplat.py:140-192,
costvol.py:4-43) and the stub compiles that text with g++ behind a serial ``__global__`` / ``atomicAdd`` shim, so
``softsplat_func`` / ``softsplat`` / ``costvol_func`` below ARE the reference's, launch code included.  Checked, bit for bit:
  1. the plain-C restatement (oracle/m2m_ops.c) vs the reference ops on edge-case inputs (out-of-range / integer /
     non-finite flows, N>1, C=1..4, the "soft" wrapper) -> also written as tests/golden/m2m_ops_ref.npz;
  2. oracle/m2m_model_oracle.py vs the reference's M2M_PWC running on its own ops.
Writes oracle/VALIDATION_M2M.log."""
import importlib.util
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import m2m_spec, synth  # noqa: E402
from oracle import m2m_model_oracle as MO, m2m_oracle, ref_import  # noqa: E402


def load_m2m_arch():
    ops = ref_import.reference_ops()          # the reference's own vfi_models.ops on the host shim
    spec = importlib.util.spec_from_file_location("M2M_arch", os.path.join(ref_import.REFERENCE, "vfi_models/m2m/M2M_arch.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert m.softsplat_func is ops.softsplat_func and m.costvol_func is ops.costvol_func
    return m, ops


def ops_cases():
    """Inputs for the two custom ops, edge cases included; deterministic."""
    g = torch.Generator().manual_seed(7)
    cases = {}
    # splat: name -> (in [N,C,H,W], flow [N,2,H,W])
    x = torch.rand(2, 4, 24, 40, generator=g)
    fl = (torch.rand(2, 2, 24, 40, generator=g) - 0.5) * 20
    cases["splat_random10px"] = (x, fl)
    fl2 = (torch.rand(2, 2, 24, 40, generator=g) - 0.5) * 120           # most targets leave the image
    cases["splat_out_of_range"] = (x, fl2)
    fl3 = torch.randint(-3, 4, (2, 2, 24, 40), generator=g).float()      # integer flows: weights exactly 1,0,0,0
    cases["splat_integer"] = (x, fl3)
    fl4 = fl.clone()
    fl4[0, 0, 3, 5], fl4[0, 1, 7, 9], fl4[1, 0, 0, 0], fl4[1, 1, 23, 39] = float("nan"), float("inf"), float("-inf"), float("nan")
    cases["splat_nonfinite"] = (x, fl4)
    cases["splat_c1"] = (torch.rand(1, 1, 17, 23, generator=g), (torch.rand(1, 2, 17, 23, generator=g) - 0.5) * 6)
    cases["splat_c3_smooth"] = (torch.rand(2, 3, 33, 47, generator=g),
                                torch.nn.functional.interpolate((torch.rand(2, 2, 5, 6, generator=g) - 0.5) * 16, size=(33, 47),
                                                                mode="bilinear", align_corners=True))
    cases["splat_zero_flow"] = (x, torch.zeros(2, 2, 24, 40))
    # cost volume: name -> (one, two) [N,32,H,W]
    a, b = torch.randn(2, 32, 12, 20, generator=g), torch.randn(2, 32, 12, 20, generator=g)
    cases["costvol_random"] = (a, b)
    cases["costvol_tiny"] = (torch.randn(1, 32, 3, 5, generator=g), torch.randn(1, 32, 3, 5, generator=g))   # smaller than the 9x9 window
    cases["costvol_same"] = (a, a.clone())
    return cases


def check_ops(ops, log):
    from oracle import gmfss_oracle

    ok = True
    golden = {}
    for name, (p, q) in ops_cases().items():
        if name.startswith("splat"):
            ref = ops.softsplat_func.apply(p, q).numpy()
            mine = m2m_oracle.softsplat_sum(p.numpy(), q.numpy())
        else:
            ref = ops.costvol_func.apply(p, q).numpy()
            mine = m2m_oracle.costvol(p.numpy(), q.numpy())
        same = bool((ref == mine).all()) and bool(np.isfinite(ref).all())
        log(f"{name}: reference kernel (host-compiled) vs oracle/m2m_ops.c: {'bit-exact' if same else 'DIFFERENT'}; "
            f"sum {float(ref.astype(np.float64).sum()):.6f}, nonzero {int((ref != 0).sum())}/{ref.size}")
        ok &= same
        golden[name + "_a"], golden[name + "_b"], golden[name + "_out"] = p.numpy(), q.numpy(), ref
    # the "soft" wrapper (softsplat.py:382-435) as GMFSS uses it, on the reference's op vs the oracle's restatement
    p, q = ops_cases()["splat_c3_smooth"]
    metric = torch.randn(2, 1, 33, 47, generator=torch.Generator().manual_seed(9))
    ref = ops.softsplat(p, q, metric, "soft").numpy()
    mine = gmfss_oracle.softsplat_soft(p, q, metric).numpy()
    same = bool((ref == mine).all())
    log(f"softsplat(..., 'soft') wrapper: reference vs oracle/gmfss_oracle.softsplat_soft: {'bit-exact' if same else 'DIFFERENT'}")
    ok &= same
    golden["soft_in"], golden["soft_flow"], golden["soft_metric"], golden["soft_out"] = p.numpy(), q.numpy(), metric.numpy(), ref
    # the float overload of abs() in the host build (a silent int abs would truncate): |0.25 - 0.75| averaged
    one, two = torch.full((1, 32, 3, 5), 0.25), torch.full((1, 32, 3, 5), 0.75)
    cv = ops.costvol_func.apply(one, two)
    assert cv[0, 40, 1, 2].item() == 0.5 and cv[0, 0, 0, 0].item() == 0.25, "host shim resolved abs() to the int overload"
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "m2m_ops_ref.npz"), **golden)
    log(f"wrote tests/golden/m2m_ops_ref.npz ({len(golden)} arrays): outputs of the reference's own kernels")
    return ok


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    m, ops = load_m2m_arch()
    ok_ops = check_ops(ops, log)
    sd = synth.m2m_synth_state_dict(1234)
    net = m.M2M_PWC()
    assert list(net.state_dict().keys()) == list(m2m_spec.m2m_shapes().keys()), "key order differs"
    net.load_state_dict(sd, strict=True)
    net.eval()
    log(f"reference M2M_PWC loaded synthetic state_dict strictly: {len(sd)} tensors, {sum(v.numel() for v in sd.values())} params")
    ok = ok_ops
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 5, 30, 44, generator=g)
    fl = (torch.rand(2, 2, 30, 44, generator=g) - 0.5) * 30
    d = (m.backwarp(x, fl) - MO.backwarp(x, fl)).abs().max().item()
    log(f"backwarp (zeros, align_corners=True): max|ref-oracle| = {d:.3e}")
    ok &= d == 0.0
    for (h, w, times) in ((64, 64, [0.5]), (100, 150, [0.5, 0.25]), (270, 480, [0.5])):
        fr = synth.smooth_frames(2, h, w, seed=h, shift=3.0)
        i0 = fr[0:1].permute(0, 3, 1, 2).contiguous()
        i1 = fr[1:2].permute(0, 3, 1, 2).contiguous()
        ts = [torch.tensor([t]).view(1, 1, 1, 1) for t in times]
        with torch.inference_mode():
            t0 = time.time()
            a = net(i0, i1, ts)
            t1 = time.time()
            b, aux = MO.m2m_forward(sd, i0, i1, ts, return_aux=True)
            t2 = time.time()
        d = max((x - y).abs().max().item() for x, y in zip(a, b))
        log(f"M2M_PWC {h}x{w} t={times}: max|ref-oracle| = {d:.3e}; max|flow| pwc {aux['fwd'].abs().max().item():.2f} refined "
            f"{aux['ten_fwd'].abs().max().item():.2f}px; out range [{b[0].min().item():.3f},{b[0].max().item():.3f}] "
            f"(ref {t1 - t0:.1f}s, oracle {t2 - t1:.1f}s)")
        ok &= d == 0.0
    log("M2M VALIDATION " + ("PASSED (bit-exact; ops pinned by execution of the reference text: the reference side ran its own cupy_ops kernels, host-compiled; the oracle side ran oracle/m2m_ops.c)" if ok else "FAILED"))
    with open(os.path.join(ROOT, "oracle", "VALIDATION_M2M.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
