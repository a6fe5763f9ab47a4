import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
ge.build(); ge.load_package()
from cfi_amd import synth
from cfi_amd.ifunet import IFUNetEngine
sd = synth.ifunet_synth_state_dict(1234)
fr = synth.smooth_frames(2, 1080, 1920, seed=2, shift=4.0)
x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
outs = {}
for mode in (True, False, True, False):
    eng = IFUNetEngine(sd); eng.fork_stages = mode
    out = torch.empty(1080, 1920, 3, device="cuda")
    for _ in range(3): eng.forward(x0, x1, 0.5, out, scale=1.0, ensemble=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): eng.forward(x0, x1, 0.5, out, scale=1.0, ensemble=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    outs.setdefault(mode, out.clone())
    print(f"fork_stages={mode}: {dt*1e3:.2f} ms per frame; graphs {[type(g).__name__ for g in eng._graphs.values()]}", flush=True)
    eng.close()
print("identical:", torch.equal(outs[True], outs[False]))
