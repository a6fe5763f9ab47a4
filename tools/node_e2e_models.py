"""End-to-end node timing (host tensor in, host tensor out) for the FILM and M2M nodes at 1080p
(BASELINE.json configs[2] N=5 and configs[4] N=9), as ComfyUI would call them."""
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
from cfi_amd import film, hostpipe, m2m, synth  # noqa: E402

if __name__ == "__main__":
    # usage: node_e2e_models.py [frames of the M2M clip] [frames of the FILM clip]   (defaults 9 / 5 = BASELINE.json configs[4] / configs[2])
    n_m2m = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    n_film = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    base = synth.smooth_frames(3, 1080, 1920, seed=1, shift=4.0)
    with tempfile.TemporaryDirectory() as td:
        for name, mod, cls, sd, n, mults in (("M2M", m2m, "M2M_VFI", synth.m2m_synth_state_dict(1234), n_m2m, (2, 4)),
                                             ("FILM", film, "FILM_VFI", synth.film_synth_state_dict(1234), n_film, (2,))):
            pth = os.path.join(td, name + ".pth")
            torch.save(sd, pth)
            mod.load_file_from_github_release = lambda model_type, ckpt, p=pth: p
            frames = base[torch.arange(n) % 3].contiguous()
            node = getattr(mod, cls)()
            node.vfi("x", frames[:2], multiplier=2)   # warm-up
            for m in mults:
                for rep in range(2):
                    hostpipe.stats.clear()
                    t0 = time.perf_counter()
                    res = node.vfi("x", frames, multiplier=m)
                    dt = time.perf_counter() - t0
                    if hostpipe.PROFILE and rep == 1:      # VFI_HOST_PROFILE=1: seconds summed over worker / main-thread calls
                        print("   host phases: " + ", ".join(f"{k} {v[0]}x {v[1] * 1e3:.1f} ms" for k, v in sorted(hostpipe.stats.items())), flush=True)
                    out = res[0]
                    del res
                    new = out.shape[0] - n
                    print(f"{name} node e2e: {n} frames 1080p x{m} -> {out.shape[0]} frames: {dt:.3f} s, {new / dt:.1f} interpolated frames/s "
                          f"(engine cached across calls; VFI_MODEL_CACHE=0 reloads per call like the reference)", flush=True)
