"""ORACLE tooling — pin oracle/film_oracle.py against the reference's film_arch.Interpolator, here, on CPU.

    python oracle/validate_film_vs_reference.py

(The FILM node itself executes a TorchScript artifact that is not in the tree; film_arch.py is its source mirror.)
Appends to oracle/VALIDATION_FILM.log."""
import importlib.util
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import film_spec, synth  # noqa: E402
from oracle import film_oracle, ref_import  # noqa: E402


def load_film_arch():
    ref_import.setup()
    spec = importlib.util.spec_from_file_location("film_arch", os.path.join(ref_import.REFERENCE, "vfi_models/film/film_arch.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    m = load_film_arch()
    sd = synth.film_synth_state_dict(1234)
    net = m.Interpolator()
    assert list(net.state_dict().keys()) == list(film_spec.film_shapes().keys()), "key order differs"
    net.load_state_dict(sd, strict=True)
    net.eval()
    log(f"reference film_arch.Interpolator loaded synthetic state_dict strictly: {len(sd)} tensors, "
        f"{sum(v.numel() for v in sd.values())} params")
    ok = True
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1, 6, 40, 56, generator=g)
    fl = (torch.rand(1, 2, 40, 56, generator=g) - 0.5) * 40
    d = (m.warp(x, fl) - film_oracle.warp(x, fl)).abs().max().item()
    log(f"warp (align_corners=False, border) max|ref-oracle| = {d:.3e}")
    ok &= d == 0.0
    for (h, w) in ((64, 96), (135, 240), (270, 480)):
        fr = synth.smooth_frames(2, h, w, seed=h, shift=2.0)
        x0 = fr[0:1].permute(0, 3, 1, 2).contiguous()
        x1 = fr[1:2].permute(0, 3, 1, 2).contiguous()
        with torch.inference_mode():
            t0 = time.time()
            a = net(x0, x1, torch.full((1, 1), .5))
            t1 = time.time()
            b, aux = film_oracle.film_forward(sd, x0, x1, return_aux=True)
            t2 = time.time()
        d = (a - b).abs().max().item()
        log(f"Interpolator {h}x{w}: max|ref-oracle| = {d:.3e}; max|flow| = {aux['fwd_flow'][0].abs().max().item():.2f}px "
            f"out range [{b.min().item():.3f},{b.max().item():.3f}] (ref {t1 - t0:.1f}s, oracle {t2 - t1:.1f}s)")
        ok &= d == 0.0
    # bisection schedule of the node (film/__init__.py:12-42) with a dummy midpoint model
    import vfi_models.film as FN

    class Mid(torch.nn.Module):
        def forward(self, a, b, dt):
            return (a + b) * 0.5

    for mult in (2, 3, 4, 5, 8):
        a = torch.zeros(1, 3, 4, 4)
        b = torch.ones(1, 3, 4, 4)
        r = FN.inference(Mid(), a, b, mult - 1)
        ref_pos = [float(t[0, 0, 0, 0]) for t in r]
        frames = torch.stack([a[0].permute(1, 2, 0), b[0].permute(1, 2, 0)])
        mine = film_oracle.film_vfi(None, frames, multiplier=mult, model=lambda p, q: (p + q) * 0.5)
        my_pos = [float(v) for v in mine[:, 0, 0, 0]]
        same = ref_pos == my_pos
        log(f"schedule x{mult}: reference positions {ref_pos} equal={same}")
        ok &= same
    log("FILM VALIDATION " + ("PASSED (bit-exact)" if ok else "FAILED"))
    with open(os.path.join(ROOT, "oracle", "VALIDATION_FILM.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
