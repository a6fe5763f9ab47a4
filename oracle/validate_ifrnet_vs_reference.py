"""ORACLE tooling — pin oracle/ifrnet_oracle.py against the reference's own IRFNet_L / IRFNet_S modules and the IFRNet_VFI
node (vfi_models/ifrnet), here, on CPU, with seeded synthetic checkpoints; write tests/golden/ifrnet_*.npz (outputs of
the REFERENCE).  Bit-exact agreement is required.  Writes oracle/VALIDATION_IFRNET.log."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import ifrnet_spec, synth  # noqa: E402
from cfi_amd.schedule import InterpolationStateList  # noqa: E402
from oracle import ifrnet_oracle, ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    ref_import.setup()
    from vfi_models.ifrnet.IFRNet_L_arch import IRFNet_L
    from vfi_models.ifrnet.IFRNet_S_arch import IRFNet_S
    import vfi_models.ifrnet as N

    ok = True
    golden = {}
    for kind, cls in (("L", IRFNet_L), ("S", IRFNet_S)):
        sd = synth.ifrnet_synth_state_dict(kind, 1234)
        net = cls()
        assert list(net.state_dict().keys()) == list(ifrnet_spec.ifrnet_shapes(kind).keys()), f"IFRNet_{kind}: key order differs"
        net.load_state_dict(sd, strict=True)
        net.eval()
        log(f"IFRNet_{kind}: reference module loaded the synthetic state_dict strictly: {len(sd)} tensors, "
            f"{sum(v.numel() for v in sd.values())} params")
        for (h, w, bsz, sf, t) in [(64, 64, 1, 1.0, 0.5), (100, 150, 2, 0.5, 1.0), (128, 192, 1, 0.25, 0.3), (200, 328, 1, 0.5, 1.0),
                                   (90, 70, 1, 1.0, 0.25)]:
            fr = synth.smooth_frames(2, h, w, seed=h + 1, shift=2.5)
            i0 = fr[0:1].permute(0, 3, 1, 2).repeat(bsz, 1, 1, 1).contiguous()
            i1 = fr[1:2].permute(0, 3, 1, 2).repeat(bsz, 1, 1, 1).contiguous()
            with torch.inference_mode():
                r = net(i0, i1, sf, t)
                o, aux = ifrnet_oracle.ifrnet_forward(sd, i0, i1, sf, t, return_aux=True)
            d = (r - o).abs().max().item()
            fmax = max(a.abs().max().item() for a in aux["final_flow"])
            log(f"IFRNet_{kind} forward {h}x{w} B={bsz} scale_factor={sf} timestep={t}: max|ref-oracle| = {d:.3e}   "
                f"max|final flow| = {fmax:.2f} px")
            ok &= d == 0.0
        # the node, through the reference's generic_frame_loop
        with tempfile.TemporaryDirectory() as td:
            pth = os.path.join(td, f"IFRNet_{kind}_Vimeo90K.pth")
            torch.save(sd, pth)
            N.load_file_from_github_release = lambda model_type, ckpt: pth
            frames = synth.smooth_frames(3, 72, 100, seed=11, shift=3.0)
            golden[f"{kind}_frames"] = frames.numpy()
            for name, kw in (("x2", dict(multiplier=2)), ("x2_t05", dict(multiplier=2, scale_factor=0.5)),
                             ("x4_skip1", dict(multiplier=4, optional_interpolation_states=InterpolationStateList([1], True)))):
                try:
                    with torch.inference_mode():
                        (r,) = N.IFRNet_VFI().vfi(os.path.basename(pth), frames.clone(), clear_cache_after_n_frames=10, **kw)
                except Exception as e:   # noqa: BLE001 — record what the reference does
                    log(f"IFRNet_{kind} node {name}: the reference raises {type(e).__name__}: {str(e)[:160]}")
                    continue
                okw = dict(kw)
                states = okw.pop("optional_interpolation_states", None)
                o = ifrnet_oracle.ifrnet_vfi(sd, frames, states=states, **okw)
                same = r.shape == o.shape and (r - o).abs().max().item() == 0.0
                log(f"IFRNet_{kind} node {name}: out {tuple(r.shape)} max|ref-oracle| = "
                    f"{(r - o).abs().max().item() if r.shape == o.shape else float('nan'):.3e}")
                ok &= same
                golden[f"{kind}_{name}"] = r.numpy()
    log("RESULT: " + ("oracle == reference, bit-exact on every case" if ok else "MISMATCH"))
    np.savez_compressed(os.path.join(OUT, "ifrnet_node.npz"), **golden)
    log(f"wrote tests/golden/ifrnet_node.npz ({os.path.getsize(os.path.join(OUT, 'ifrnet_node.npz')) / 1e6:.2f} MB): "
        + ", ".join(f"{k}{tuple(v.shape)}" for k, v in golden.items()))
    with open(os.path.join(ROOT, "oracle", "VALIDATION_IFRNET.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
