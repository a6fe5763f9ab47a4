"""End-to-end node timing (host tensor in, host tensor out) of the op-by-op nodes — IFUnet VFI, GMFSS Fortuna VFI — at 1080p, call after call
(engines, workspaces and captured graphs stay between calls of one frame shape: ckpt.end_call).   usage: node_e2e_ops.py [frames] [ifunet|gmfss]"""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.build()
ge.load_package()
from cfi_amd import ckpt, gmfss, ifunet, synth  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
    which = sys.argv[2:] or ["ifunet", "gmfss"]
    with tempfile.TemporaryDirectory() as td:
        for name in which:
            if name == "ifunet":
                pth = os.path.join(td, "IFUNet.pth")
                torch.save(synth.ifunet_synth_state_dict(1234), pth)
                ckpt.load_file_from_github_release = lambda model_type, ck, p=pth: p
                node, arg = ifunet.IFUnet_VFI(), "IFUNet.pth"
                base = synth.smooth_frames(3, 1080, 1920, seed=1, shift=4.0)
            else:
                sds = synth.gmfss_coherent_state_dicts(1234, "union")
                paths = {}
                for part, (_, fn) in gmfss.CKPTS_PATH_CONFIG["GMFSS_fortuna_union"].items():
                    paths[fn] = os.path.join(td, fn)
                    torch.save(sds[part], paths[fn])
                ckpt.load_file_from_github_release = lambda model_type, ck, p=paths: p[ck]
                node, arg = gmfss.GMFSS_Fortuna_VFI(), "GMFSS_fortuna_union"
                base = synth.texture_frames(4, 1080, 1920, seed=2, cell=16)[:3].contiguous()
            frames = base[torch.arange(n) % 3].contiguous()
            for rep in range(4):
                t0 = time.perf_counter()
                (out,) = node.vfi(arg, frames, multiplier=2)
                dt = time.perf_counter() - t0
                print(f"{name} node e2e, call {rep}: {n} frames 1080p x2 -> {out.shape[0]} frames: {dt:.3f} s, {(out.shape[0] - n) / dt:.1f} interpolated frames/s "
                      f"(VFI_PAIR_LANES={os.environ.get('VFI_PAIR_LANES', 'default')})", flush=True)
                del out
            ckpt.clear_engine_cache()
