"""ORACLE tooling — generate tests/golden/*.npz by executing the REAL reference on CPU.

    python oracle/make_golden.py

Runs only in the build container (needs /root/reference).  The vectors are small, committed, and
re-checked by tests/test_oracle_golden.py (oracle vs reference output) and tests/test_gpu_rife.py
(HIP path vs reference output) on boxes where the reference is absent.  Weights are the
deterministic synthetic checkpoint ``synth.rife47_synth_state_dict(1234)`` loaded into the
reference's own ``IFNet("4.7")`` with ``strict=True``.
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pkgload import load_package  # noqa: E402

load_package()
from cfi_amd import synth  # noqa: E402
from cfi_amd.schedule import InterpolationStateList  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def demo_pair(name0, name1, y0, x0, h, w):
    from PIL import Image

    out = []
    for n in (name0, name1):
        im = np.asarray(Image.open(os.path.join(ref_import.REFERENCE, "demo_frames", n)).convert("RGB"))
        out.append(torch.from_numpy(im[y0 : y0 + h, x0 : x0 + w].astype(np.float32) / 255.0))
    return torch.stack(out)  # [2,h,w,3]


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = ref_import.rife_arch()
    sd = synth.rife47_synth_state_dict(1234)
    net = ref.IFNet("4.7")
    net.load_state_dict(sd, strict=True)
    net.eval()

    # ---- warp (rife_arch.py:31-70): in-range and far out-of-range flows
    g = torch.Generator().manual_seed(21)
    x = torch.rand(2, 4, 40, 56, generator=g)
    fl = (torch.rand(2, 2, 40, 56, generator=g) - 0.5) * 30
    fl[:, :, :4] *= 8  # rows that leave the image
    y = ref.warp(x, fl)
    np.savez_compressed(os.path.join(OUT, "rife_warp.npz"), x=x.numpy(), flow=fl.numpy(), y=y.numpy())

    # ---- IFNet 4.7 forward: real-image crop (anime demo frames), odd size -> exercises pad/crop
    fr = demo_pair("anime0.png", "anime1.png", 180, 380, 100, 150)
    i0 = fr[0:1].permute(0, 3, 1, 2).repeat(2, 1, 1, 1).contiguous()
    i1 = fr[1:2].permute(0, 3, 1, 2).repeat(2, 1, 1, 1).contiguous()
    ts = torch.tensor([0.5, 0.25]).view(-1, 1, 1, 1)
    with torch.inference_mode():
        out = net(i0, i1, ts, [8, 4, 2, 1], False, False)
    np.savez_compressed(os.path.join(OUT, "rife47_net_anime.npz"), frames=fr.numpy(), timesteps=ts.view(-1).numpy(),
                        out=out.permute(0, 2, 3, 1).contiguous().numpy())

    # ---- the node end to end (scheduling, alpha drop, interleave, clamp); 4-channel input
    frames = synth.smooth_frames(5, 50, 70, seed=5, shift=2.0, c=4)
    cases = {
        "m2": dict(multiplier=2),
        "m3_bs2": dict(multiplier=3, batch_size=2),
        "mlist": dict(multiplier=[3, 0, 1]),
        "m2_skip12": dict(multiplier=2, optional_interpolation_states=InterpolationStateList([1, 2], True)),
        "m2_keep12": dict(multiplier=2, optional_interpolation_states=InterpolationStateList([1, 2], False)),
    }
    node_out = {}
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "rife47.pth")
        torch.save(sd, pth)
        R = ref_import.rife_node(pth)
        for name, kw in cases.items():
            R._model_cache.clear()
            node_out[name] = R.RIFE_VFI().vfi("rife47.pth", frames, **kw)[0].numpy()

        # ---- schedule known answers with a dummy linear-blend model: positions in time
        class Dummy(torch.nn.Module):
            def forward(self, f0, f1, t, *a):
                return f0 * (1 - t) + f1 * t

        tf = (torch.arange(5, dtype=torch.float32) / 4).view(5, 1, 1, 1).expand(5, 4, 4, 3).contiguous()  # in [0,1]: the node clamps
        kat = {}
        for name, kw in cases.items():
            R._model_cache.clear()
            R._model_cache[("rife47.pth", "float32", False)] = Dummy()
            o = R.RIFE_VFI().vfi("rife47.pth", tf, **kw)[0]
            kat[name] = [round(float(v) * 4, 5) for v in o[:, 0, 0, 0]]
        # ---- BASELINE.json configs[0]: the node on the full demo_frames/anime0+anime1 pair (540x960), 2x.
        # Inputs are stored as the PNG's uint8 pixels (frames = u8 / 255), the output as the synthesised middle frame.
        from PIL import Image
        u8 = np.stack([np.asarray(Image.open(os.path.join(ref_import.REFERENCE, "demo_frames", n)).convert("RGB"))
                       for n in ("anime0.png", "anime1.png")])
        fr = torch.from_numpy(u8.astype(np.float32) / 255.0)
        R._model_cache.clear()
        o = R.RIFE_VFI().vfi("rife47.pth", fr, multiplier=2)[0]
        assert o.shape[0] == 3 and torch.equal(o[0], fr[0]) and torch.equal(o[2], fr[1])
        np.savez_compressed(os.path.join(OUT, "rife47_node_anime540.npz"), frames_u8=u8, mid=o[1].numpy())
    np.savez_compressed(os.path.join(OUT, "rife47_node.npz"), frames=frames.numpy(), **node_out)
    with open(os.path.join(OUT, "rife_schedule_kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
