"""GMFSS at 1080p: the union head's IFNet 4.6 pass on the engine's side stream beside the splats (GMFSSEngine.fork_stages) vs one stream,
same process."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build(); ge.load_package()
from cfi_amd import synth
from cfi_amd.gmfss import GMFSSEngine
sds = synth.gmfss_coherent_state_dicts(1234, "union")
fr = synth.texture_frames(4, 1080, 1920, seed=2, cell=16)[:2].contiguous()
x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
outs = {}
for mode in (True, False, True, False):
    eng = GMFSSEngine(sds); eng.fork_stages = mode
    out = torch.empty(1080, 1920, 3, device="cuda")
    for _ in range(3): eng.prepare(x0, x1); eng.render(0.5, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): eng.prepare(x0, x1); eng.render(0.5, out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    outs.setdefault(mode, out.clone())
    print(f"fork_stages={mode}: {dt*1e3:.2f} ms per pair (prepare + render); graphs {[type(g).__name__ for g in eng._graphs.values()]}; workspace {eng.workspace_bytes()/2**30:.2f} GiB", flush=True)
    eng.close()
print("max |difference|:", (outs[True] - outs[False]).abs().max().item())
