"""ORACLE tooling — generate tests/golden/{film,m2m}_*.npz by executing the REAL reference modules on CPU.

    python oracle/make_golden_film_m2m.py

Runs only in the build container (needs /root/reference).  What is executed:
  * FILM: vfi_models/film/film_arch.py ``Interpolator`` (the in-tree source mirror of the TorchScript artifact the
    node loads, which is absent: parity with the artifact itself stays unpinned);
  * M2M: vfi_models/m2m/M2M_arch.py ``M2M_PWC`` and the real node ``M2M_VFI.vfi`` (vfi_models/m2m/__init__.py) through
    ``vfi_utils.generic_frame_loop``, with ``vfi_models.ops`` replaced by a stand-in that forwards the two cupy ops to
    the plain-C restatements of their kernel text (oracle/m2m_ops.c) — everything except those two ops is reference code.
Weights: ``synth.film_synth_state_dict(1234)`` / ``synth.m2m_synth_state_dict(1234)``, loaded with strict=True.
"""
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
from oracle.validate_film_vs_reference import load_film_arch  # noqa: E402
from oracle.validate_m2m_vs_reference import load_m2m_arch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

M2M_NODE_CASES = {
    "m2": dict(multiplier=2),
    "m3_skip1": dict(multiplier=3, optional_interpolation_states=InterpolationStateList([1], True)),
    "mlist_203": dict(multiplier=[2, 0, 3]),
    "mlist_120": dict(multiplier=[1, 2, 0]),
    "mlist_3_keep0": dict(multiplier=[3], optional_interpolation_states=InterpolationStateList([0], False)),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    # ---- FILM Interpolator
    fa = load_film_arch()
    sd = synth.film_synth_state_dict(1234)
    net = fa.Interpolator()
    net.load_state_dict(sd, strict=True)
    net.eval()
    fr = synth.smooth_frames(2, 64, 96, seed=11, shift=3.0)
    x = fr.permute(0, 3, 1, 2).contiguous()
    with torch.inference_mode():
        out = net(x[0:1], x[1:2], torch.full((1, 1), 0.5))
    np.savez_compressed(os.path.join(OUT, "film_net.npz"), frames=fr.numpy(), out=out.permute(0, 2, 3, 1).contiguous().numpy())

    # ---- M2M model
    ma, _ops = load_m2m_arch()          # the reference's M2M_arch on the reference's own (host-compiled) cupy_ops
    sd = synth.m2m_synth_state_dict(1234)
    net = ma.M2M_PWC()
    net.load_state_dict(sd, strict=True)
    net.eval()
    fr = synth.smooth_frames(2, 70, 100, seed=12, shift=3.0)
    x = fr.permute(0, 3, 1, 2).contiguous()
    times = [0.5, 0.25]
    with torch.inference_mode():
        outs = net(x[0:1], x[1:2], [torch.tensor([t]).view(1, 1, 1, 1) for t in times])
    np.savez_compressed(os.path.join(OUT, "m2m_net.npz"), frames=fr.numpy(), times=np.asarray(times, np.float32),
                        out=torch.cat(outs, 0).permute(0, 2, 3, 1).contiguous().numpy())

    # ---- the real M2M node (generic_frame_loop): int / list multipliers, skip and keep lists; RGBA input
    import vfi_models.m2m as M

    frames = synth.smooth_frames(4, 64, 64, seed=13, shift=2.0, c=4)
    node_out = {}
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "M2M.pth")
        torch.save(sd, pth)
        M.load_file_from_github_release = lambda model_type, ckpt: pth
        for name, kw in M2M_NODE_CASES.items():
            node_out[name] = M.M2M_VFI().vfi("M2M.pth", frames, **kw)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "m2m_node.npz"), frames=frames.numpy(), **node_out)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
