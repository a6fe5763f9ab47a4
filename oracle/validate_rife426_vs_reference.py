"""ORACLE tooling — pin the arch "4.26" branch of oracle/rife_oracle.py (Head encoder, 5 IFBlocks, 8 carried block-feature channels) against the
reference's own IFNet("4.26") and the RIFE_VFI node with rife426.pth, here, on CPU; write tests/golden/rife426_*.npz.

    python oracle/validate_rife426_vs_reference.py

Bit-exact agreement is required.  Appends to oracle/VALIDATION.log (section "arch 4.26")."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import rife_spec, synth  # noqa: E402
from oracle import ref_import, rife_oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    ref = ref_import.rife_arch()
    sd = synth.rife426_synth_state_dict(1234)
    net = ref.IFNet("4.26")
    assert list(net.state_dict().keys()) == list(rife_spec.rife426_shapes().keys()), "key order differs"
    net.load_state_dict(sd, strict=True)
    net.eval()
    log(f"arch 4.26: reference IFNet('4.26') loaded synthetic state_dict strictly: {len(sd)} tensors, "
        f"{sum(v.numel() for v in sd.values())} params")
    ok = True
    for (h, w, bsz, scales) in ((100, 150, 2, (16, 8, 4, 2, 1)), (64, 64, 1, (16, 8, 4, 2, 1)), (270, 480, 1, (16, 8, 4, 2, 1)), (120, 200, 1, (8, 4, 2, 1, 0.5))):
        fr = synth.smooth_frames(2, h, w, seed=3, shift=2.5)
        i0 = fr[0:1].permute(0, 3, 1, 2).repeat(bsz, 1, 1, 1).contiguous()
        i1 = fr[1:2].permute(0, 3, 1, 2).repeat(bsz, 1, 1, 1).contiguous()
        ts = torch.tensor([0.5, 0.25][:bsz]).view(-1, 1, 1, 1)
        with torch.inference_mode():
            r = net(i0, i1, ts, list(scales), False, False)
            o, aux = rife_oracle.ifnet47_forward(sd, i0, i1, ts, scales, return_aux=True, arch="4.26")
        d = (r - o).abs().max().item()
        log(f"IFNet 4.26 {h}x{w} B={bsz} scales {scales}: max|ref-oracle| = {d:.3e}  max|flow| = "
            f"{max(a[0].abs().max().item() for a in aux):.2f}px")
        ok &= d == 0.0
    # goldens: the net on an anime crop, and the node with rife426.pth (RGBA clip, list multiplier)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden import demo_pair

    fr = demo_pair("anime0.png", "anime1.png", 180, 380, 100, 150)
    i0 = fr[0:1].permute(0, 3, 1, 2).repeat(2, 1, 1, 1).contiguous()
    i1 = fr[1:2].permute(0, 3, 1, 2).repeat(2, 1, 1, 1).contiguous()
    ts = torch.tensor([0.5, 0.25]).view(-1, 1, 1, 1)
    with torch.inference_mode():
        out = net(i0, i1, ts, [16, 8, 4, 2, 1], False, False)
    np.savez_compressed(os.path.join(OUT, "rife426_net_anime.npz"), frames=fr.numpy(), timesteps=ts.view(-1).numpy(),
                        out=out.permute(0, 2, 3, 1).contiguous().numpy())
    frames = synth.smooth_frames(4, 50, 70, seed=5, shift=2.0, c=4)
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "rife426.pth")
        torch.save(sd, pth)
        R = ref_import.rife_node(pth)
        node = {"m2": R.RIFE_VFI().vfi("rife426.pth", frames, multiplier=2)[0]}
        R._model_cache.clear()
        node["mlist_bs2"] = R.RIFE_VFI().vfi("rife426.pth", frames, multiplier=[3, 1], batch_size=2)[0]
    for name, kw in (("m2", dict(multiplier=2)), ("mlist_bs2", dict(multiplier=[3, 1], batch_size=2))):
        o = rife_oracle.rife_vfi(sd, frames, arch="4.26", **kw)
        d = (o - node[name]).abs().max().item()
        log(f"RIFE_VFI node rife426.pth {name}: max|ref-oracle| = {d:.3e}, {tuple(o.shape)}")
        ok &= d == 0.0 and o.shape == node[name].shape
    np.savez_compressed(os.path.join(OUT, "rife426_node.npz"), frames=frames.numpy(), **{k: v.numpy() for k, v in node.items()})
    log("RIFE 4.26 VALIDATION " + ("PASSED (bit-exact)" if ok else "FAILED"))
    log_path = os.path.join(ROOT, "oracle", "VALIDATION.log")
    prev = open(log_path).read() if os.path.exists(log_path) else ""
    marker = "---- arch 4.26 ----\n"
    prev = prev.split(marker)[0].rstrip("\n") + "\n"
    with open(log_path, "w") as f:
        f.write(prev + marker + "\n".join(lines) + "\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
