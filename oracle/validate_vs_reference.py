"""ORACLE tooling — pin oracle/rife_oracle.py against the REAL reference, here, on CPU.

    python oracle/validate_vs_reference.py [--full]

Runs the reference's own ``IFNet("4.7")`` / ``warp`` / ``RIFE_VFI.vfi`` (imported from
/root/reference through oracle/stubs) and this repo's functional restatement on the same
seeded inputs and synthetic checkpoint, and requires bit-exact agreement.  Writes the
result log to oracle/VALIDATION.log (committed).
"""
import argparse
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import synth  # noqa: E402
from oracle import ref_import, rife_oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the 1080p case")
    args = ap.parse_args()
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    ref = ref_import.rife_arch()
    sd = synth.rife47_synth_state_dict(1234)
    net = ref.IFNet("4.7")
    net.load_state_dict(sd, strict=True)  # proves key/shape table == reference loader
    net.eval()
    log(f"torch {torch.__version__}; reference IFNet('4.7') loaded synthetic state_dict strictly: "
        f"{len(sd)} tensors, {sum(v.numel() for v in sd.values())} params")

    ok = True
    # --- warp, incl. flows that leave the image
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 4, 64, 96, generator=g)
    fl = (torch.rand(2, 2, 64, 96, generator=g) - 0.5) * 80
    a = ref.warp(x, fl)
    b = rife_oracle.warp(x, fl)
    d = (a - b).abs().max().item()
    log(f"warp [2,4,64,96] flow U(-40,40): max|ref-oracle| = {d:.3e}")
    ok &= d == 0.0

    cases = [(100, 150, 2), (64, 64, 1), (270, 480, 1)]
    if args.full:
        cases.append((1080, 1920, 1))
    for (h, w, bsz) in cases:
        fr = synth.smooth_frames(2, h, w, seed=3, shift=2.5)
        i0 = fr[0:1].permute(0, 3, 1, 2).repeat(bsz, 1, 1, 1).contiguous()
        i1 = fr[1:2].permute(0, 3, 1, 2).repeat(bsz, 1, 1, 1).contiguous()
        ts = torch.tensor([0.5, 0.25][:bsz]).view(-1, 1, 1, 1)
        with torch.inference_mode():
            t0 = time.time()
            r = net(i0, i1, ts, [8, 4, 2, 1], False, False)
            t1 = time.time()
            o, aux = rife_oracle.ifnet47_forward(sd, i0, i1, ts, (8, 4, 2, 1), return_aux=True)
            t2 = time.time()
        d = (r - o).abs().max().item()
        fmax = max(a[0].abs().max().item() for a in aux)
        log(f"IFNet 4.7 {h}x{w} B={bsz}: max|ref-oracle| = {d:.3e}  max|flow| = {fmax:.2f}px "
            f"(ref {t1-t0:.2f}s, oracle {t2-t1:.2f}s)")
        ok &= d == 0.0

    # --- the node, end to end (config-1 plumbing: scheduling + interleave + clamp)
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "rife47.pth")
        torch.save(sd, pth)
        R = ref_import.rife_node(pth)
        from cfi_amd.schedule import InterpolationStateList

        fr = synth.smooth_frames(5, 72, 100, seed=5, shift=2.0, c=4)
        for kw in (dict(multiplier=2), dict(multiplier=3, batch_size=2), dict(multiplier=[3, 0, 1]),
                   dict(multiplier=2, optional_interpolation_states=InterpolationStateList([1, 2], True))):
            R._model_cache.clear()
            a = R.RIFE_VFI().vfi("rife47.pth", fr, **kw)[0]
            b = rife_oracle.rife_vfi(sd, fr, multiplier=kw["multiplier"], batch_size=kw.get("batch_size", 1),
                                     states=kw.get("optional_interpolation_states"))
            same = a.shape == b.shape and torch.equal(a, b)
            log(f"RIFE_VFI.vfi {kw if 'optional_interpolation_states' not in kw else 'multiplier=2, skip[1,2]'}: "
                f"out {tuple(a.shape)} equal={same}")
            ok &= same
    log("VALIDATION " + ("PASSED (bit-exact)" if ok else "FAILED"))
    with open(os.path.join(ROOT, "oracle", "VALIDATION.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
