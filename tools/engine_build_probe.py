"""Fixed costs of a node call for the engines the nodes rebuild per call: constructor (weight pack + upload), first pair, second pair
(graph capture for the op-by-op engines), third pair (steady)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build(); ge.load_package()
from cfi_amd import synth
H, W = 1080, 1920
out = torch.empty(H, W, 3, device="cuda")
def run(name, build, step, fr):
    x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e = build(); torch.cuda.synchronize(); t1 = time.perf_counter()
        ts = []
        for k in range(4):
            step(e, x0, x1); torch.cuda.synchronize(); ts.append(time.perf_counter())
        print(f"{name} (build {rep}): constructor {1e3*(t1-t0):.0f} ms, pairs 1-4: " + ", ".join(f"{1e3*(b-a):.0f}" for a, b in zip([t1]+ts, ts)) + " ms", flush=True)
        e.close()
from cfi_amd.gmfss import GMFSSEngine
from cfi_amd.ifunet import IFUNetEngine
from cfi_amd.m2m import M2MEngine
sds = synth.gmfss_coherent_state_dicts(1234, "union")
run("gmfss", lambda: GMFSSEngine(sds), lambda e, a, b: (e.prepare(a, b), e.render(0.5, out)), synth.texture_frames(4, H, W, seed=2, cell=16))
sd = synth.ifunet_synth_state_dict(1234)
run("ifunet", lambda: IFUNetEngine(sd), lambda e, a, b: e.forward(a, b, 0.5, out, scale=1.0, ensemble=True), synth.smooth_frames(2, H, W, seed=2, shift=4.0))
sm = synth.m2m_synth_state_dict(1234)
run("m2m", lambda: M2MEngine(sm), lambda e, a, b: (e.prepare(a, b), e.render(0.5, out)), synth.smooth_frames(2, H, W, seed=2, shift=4.0))
