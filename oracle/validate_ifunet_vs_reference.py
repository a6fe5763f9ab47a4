"""ORACLE tooling — pin oracle/ifunet_oracle.py against the reference's own IFUNetModel and the IFUnet_VFI node
(vfi_models/ifunet), here, on CPU, with a seeded synthetic checkpoint; write tests/golden/ifunet_node.npz (outputs of the
REFERENCE).  Bit-exact agreement is required.  Writes oracle/VALIDATION_IFUNET.log."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import ifunet_spec, synth  # noqa: E402
from cfi_amd.schedule import InterpolationStateList  # noqa: E402
from oracle import ifunet_oracle as O, ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    lines = []

    def log(s):
        print(s, flush=True)
        lines.append(s)

    ref_import.setup()
    from vfi_models.ifunet.IFUNet_arch import IFUNetModel
    import vfi_models.ifunet as N

    sd = synth.ifunet_synth_state_dict(1234)
    net = IFUNetModel()
    ref_shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert list(ref_shapes) == list(ifunet_spec.ifunet_shapes()), "key order differs"
    assert ref_shapes == {k: tuple(v) for k, v in ifunet_spec.ifunet_shapes().items()}, "shapes differ"
    net.load_state_dict(sd, strict=True)
    net.eval()
    log(f"IFUNetModel: reference module loaded the synthetic state_dict strictly: {len(sd)} tensors, "
        f"{sum(v.numel() for v in sd.values())} params")
    ok = True
    golden = {}
    with torch.inference_mode():
        for (h, w, t, scale, ens) in [(64, 64, 0.5, 1.0, False), (100, 150, 0.25, 1.0, True), (128, 192, 0.5, 0.5, True), (72, 100, 0.75, 2.0, False)]:
            fr = synth.smooth_frames(2, h, w, seed=h + 3, shift=2.5)
            x = fr.permute(0, 3, 1, 2).contiguous()
            r = net(x[0:1], x[1:2], timestep=t, scale=scale, ensemble=ens)
            o = O.ifunet_forward(sd, x[0:1], x[1:2], t, scale, ens)
            d = (r - o).abs().max().item()
            ph, pw = ((h - 1) // 64 + 1) * 64, ((w - 1) // 64 + 1) * 64
            i0 = torch.nn.functional.pad(x[0:1], (0, pw - w, 0, ph - h))
            i1 = torch.nn.functional.pad(x[1:2], (0, pw - w, 0, ph - h))
            flow, _, _ = O.ifunet(sd, torch.cat((i0, i1), 1), scale, t, ens)
            log(f"IFUNetModel {h}x{w} t={t} scale={scale} ensemble={ens}: max|ref-oracle| = {d:.3e}   max|flow| = {flow.abs().max().item():.2f} px   "
                f"out std {r.std().item():.3f}")
            ok &= d == 0.0
        with tempfile.TemporaryDirectory() as td:
            pth = os.path.join(td, "IFUNet.pth")
            torch.save(sd, pth)
            N.load_file_from_github_release = lambda model_type, ckpt: pth
            frames = synth.smooth_frames(3, 72, 100, seed=11, shift=3.0)
            golden["frames"] = frames.numpy()
            for name, kw in (("x2", dict(multiplier=2)), ("x2_noens_s05", dict(multiplier=2, scale_factor=0.5, ensemble=False)),
                             ("x3_skip0", dict(multiplier=3, optional_interpolation_states=InterpolationStateList([0], True)))):
                (r,) = N.IFUnet_VFI().vfi("IFUNet.pth", frames.clone(), clear_cache_after_n_frames=10, **kw)
                okw = dict(kw)
                states = okw.pop("optional_interpolation_states", None)
                o = O.ifunet_vfi(sd, frames, states=states, **okw)
                d = (r - o).abs().max().item() if r.shape == o.shape else float("nan")
                log(f"IFUnet_VFI node {name}: out {tuple(r.shape)} max|ref-oracle| = {d:.3e}")
                ok &= d == 0.0
                golden[name] = r.numpy()
    log("RESULT: " + ("oracle == reference, bit-exact on every case" if ok else "MISMATCH"))
    np.savez_compressed(os.path.join(OUT, "ifunet_node.npz"), **golden)
    log(f"wrote tests/golden/ifunet_node.npz ({os.path.getsize(os.path.join(OUT, 'ifunet_node.npz')) / 1e6:.2f} MB)")
    with open(os.path.join(ROOT, "oracle", "VALIDATION_IFUNET.log"), "w") as f:
        f.write("\n".join(lines) + "\n")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
