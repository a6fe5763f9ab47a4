"""Cost of a model's first pair after release_workspace() (the node releases the activations at the end of every call)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build(); ge.load_package()
from cfi_amd import synth
from cfi_amd.m2m import M2MEngine
from cfi_amd.film import FilmEngine
fr = synth.smooth_frames(2, 1080, 1920, seed=2, shift=4.0)
x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
for name, eng, step in (("m2m", M2MEngine(synth.m2m_synth_state_dict(1234)), lambda e: (e.prepare(x0, x1), e.render(0.5))),
                        ("film", FilmEngine(synth.film_synth_state_dict(1234)), lambda e: e.forward(x0, x1))):
    step(eng); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); step(eng); torch.cuda.synchronize(); warm = time.perf_counter() - t0
        t0 = time.perf_counter(); eng.release_workspace(); rel = time.perf_counter() - t0
        t0 = time.perf_counter(); step(eng); t_issue = time.perf_counter() - t0; torch.cuda.synchronize(); cold = time.perf_counter() - t0
        print(f"{name}: warm pair {warm*1e3:.1f} ms, release_workspace {rel*1e3:.1f} ms, first pair after release {cold*1e3:.1f} ms (host returns after {t_issue*1e3:.1f} ms)", flush=True)
