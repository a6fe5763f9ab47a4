"""ORACLE tooling — pin oracle/m2m_model_oracle.py against the reference's M2M_arch.M2M_PWC, here, on CPU.

The reference module imports vfi_models.ops (CuPy/Taichi, neither usable here); a stand-in module forwards
softsplat_func / costvol_func to the plain-C restatements of their CUDA kernel text (oracle/m2m_ops.c), so this pins
everything EXCEPT those two ops.  Writes oracle/VALIDATION_M2M.log."""
import importlib.util
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import m2m_spec, synth  # noqa: E402
from oracle import m2m_model_oracle as MO, m2m_oracle, ref_import  # noqa: E402


def load_m2m_arch():
    ref_import.setup()
    ops = types.ModuleType("vfi_models.ops")

    class _S:
        @staticmethod
        def apply(a, f):
            return torch.from_numpy(m2m_oracle.softsplat_sum(a.detach().numpy(), f.detach().numpy()))

    class _C:
        @staticmethod
        def apply(a, b):
            return torch.from_numpy(m2m_oracle.costvol(a.detach().numpy(), b.detach().numpy()))

    ops.softsplat_func, ops.costvol_func = _S, _C
    sys.modules["vfi_models.ops"] = ops
    spec = importlib.util.spec_from_file_location("M2M_arch", os.path.join(ref_import.REFERENCE, "vfi_models/m2m/M2M_arch.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    m = load_m2m_arch()
    sd = synth.m2m_synth_state_dict(1234)
    net = m.M2M_PWC()
    assert list(net.state_dict().keys()) == list(m2m_spec.m2m_shapes().keys()), "key order differs"
    net.load_state_dict(sd, strict=True)
    net.eval()
    log(f"reference M2M_PWC loaded synthetic state_dict strictly: {len(sd)} tensors, {sum(v.numel() for v in sd.values())} params")
    ok = True
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
    log("M2M VALIDATION " + ("PASSED (bit-exact; custom ops via the C restatement on both sides)" if ok else "FAILED"))
    with open(os.path.join(ROOT, "oracle", "VALIDATION_M2M.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
