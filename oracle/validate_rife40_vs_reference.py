"""ORACLE tooling — pin the arch "4.0" branch of oracle/rife_oracle.py (ifnet40_forward: PReLU IFBlocks, accumulated mask,
in-place scale doubling, Contextnet + Unet refinement) against the reference's own IFNet("4.0") and the RIFE_VFI node with
sudo_rife4_269.662_testV1_scale1.pth, here, on CPU; write tests/golden/rife40_*.npz.  Bit-exact agreement is required.
Appends to oracle/VALIDATION.log (section "arch 4.0")."""
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
CKPT = "sudo_rife4_269.662_testV1_scale1.pth"


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    ref = ref_import.rife_arch()
    sd = synth.rife40_synth_state_dict(1234)
    net = ref.IFNet("4.0")
    assert list(net.state_dict().keys()) == list(rife_spec.rife40_shapes().keys()), "key order differs"
    net.load_state_dict(sd, strict=True)
    net.eval()
    log(f"arch 4.0: reference IFNet('4.0') loaded synthetic state_dict strictly: {len(sd)} tensors, "
        f"{sum(v.numel() for v in sd.values())} params")
    ok = True
    big = {k: (v * 24.0 if "lastconv" in k else v) for k, v in sd.items()}   # flows above 32 px: the scale-doubling branch
    netb = ref.IFNet("4.0")
    netb.load_state_dict(big, strict=True)
    netb.eval()
    cases = [(sd, net, 100, 150, 2, True, True), (sd, net, 64, 64, 1, False, False), (sd, net, 270, 480, 1, True, False),
             (sd, net, 120, 200, 1, False, True), (big, netb, 128, 192, 1, False, False), (big, netb, 128, 192, 1, True, False)]
    for (w_, n_, h, w, bsz, training, fastmode) in cases:
        fr = synth.smooth_frames(2, h, w, seed=3, shift=2.5)
        i0 = fr[0:1].permute(0, 3, 1, 2).repeat(bsz, 1, 1, 1).contiguous()
        i1 = fr[1:2].permute(0, 3, 1, 2).repeat(bsz, 1, 1, 1).contiguous()
        ts = torch.tensor([0.5, 0.25][:bsz]).view(-1, 1, 1, 1)
        sl_ref, sl_or = [8.0, 4.0, 2.0, 1.0], [8.0, 4.0, 2.0, 1.0]
        with torch.inference_mode():
            r = n_(i0, i1, ts, sl_ref, training, fastmode)
            o, aux = rife_oracle.ifnet40_forward(w_, i0, i1, ts, sl_or, training, fastmode, return_aux=True)
        d = (r - o).abs().max().item()
        log(f"IFNet 4.0 {h}x{w} B={bsz} training={training} fastmode={fastmode}: max|ref-oracle| = {d:.3e}  max|flow| = "
            f"{max(a[0].abs().max().item() for a in aux):.2f}px  scale_list after: ref {sl_ref} oracle {sl_or}")
        ok &= d == 0.0 and sl_ref == sl_or
    # goldens
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden import demo_pair

    fr = demo_pair("anime0.png", "anime1.png", 180, 380, 100, 150)
    i0 = fr[0:1].permute(0, 3, 1, 2).repeat(2, 1, 1, 1).contiguous()
    i1 = fr[1:2].permute(0, 3, 1, 2).repeat(2, 1, 1, 1).contiguous()
    ts = torch.tensor([0.5, 0.25]).view(-1, 1, 1, 1)
    with torch.inference_mode():
        out_fast = net(i0, i1, ts, [8.0, 4.0, 2.0, 1.0], True, True)
        out_full = net(i0, i1, ts, [8.0, 4.0, 2.0, 1.0], False, False)
    np.savez_compressed(os.path.join(OUT, "rife40_net_anime.npz"), frames=fr.numpy(), timesteps=ts.view(-1).numpy(),
                        out_fast=out_fast.permute(0, 2, 3, 1).contiguous().numpy(), out_full=out_full.permute(0, 2, 3, 1).contiguous().numpy())
    frames = synth.smooth_frames(4, 50, 70, seed=5, shift=2.0, c=4)
    node_cases = {"default": dict(multiplier=2, fast_mode=True, ensemble=True),
                  "refine_bs2": dict(multiplier=[3, 1], batch_size=2, fast_mode=False, ensemble=False)}
    node = {}
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, CKPT)
        torch.save(sd, pth)
        R = ref_import.rife_node(pth)
        for name, kw in node_cases.items():
            R._model_cache.clear()
            node[name] = R.RIFE_VFI().vfi(CKPT, frames, **kw)[0]
            o = rife_oracle.rife_vfi(sd, frames, arch="4.0", **kw)
            d = (o - node[name]).abs().max().item()
            log(f"RIFE_VFI node {CKPT} {name}: max|ref-oracle| = {d:.3e}, {tuple(o.shape)}")
            ok &= d == 0.0 and o.shape == node[name].shape
    np.savez_compressed(os.path.join(OUT, "rife40_node.npz"), frames=frames.numpy(), **{k: v.numpy() for k, v in node.items()})
    log("RIFE 4.0 VALIDATION " + ("PASSED (bit-exact)" if ok else "FAILED"))
    log_path = os.path.join(ROOT, "oracle", "VALIDATION.log")
    prev = open(log_path).read() if os.path.exists(log_path) else ""
    marker = "---- arch 4.0 ----\n"
    prev = prev.split(marker)[0].rstrip("\n") + "\n"
    with open(log_path, "w") as f:
        f.write(prev + marker + "\n".join(lines) + "\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
