"""Diagnostic: RIFE outputs under VFI_STAGE_QUAD masks, compared bit for bit (run on the GPU box)."""
import os, re, subprocess, sys, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, "tests", "test_gpu_rife.py")).read()
snippet = re.search(r'_QUAD_SNIPPET = r"""(.*?)"""', src, re.S).group(1)
res = {}
for tag, mask in (("0a", "0"), ("0b", "0"), ("6", "6"), ("2", "2"), ("4", "4")):
    out = f"/tmp/quad_{tag}.pt"
    subprocess.run([sys.executable, "-c", snippet.format(root=root, out=out)], check=True, env=dict(os.environ, VFI_STAGE_QUAD=mask), timeout=300)
    res[tag] = torch.load(out)
for tag in ("0b", "6", "2", "4"):
    for i, (a, b) in enumerate(zip(res["0a"], res[tag])):
        d = (a - b).abs()
        nz = (d > 0).nonzero()
        print(tag, i, tuple(a.shape), "max", d.max().item(), "n_diff", int((d > 0).sum()), "first", nz[:3].tolist(), "last", nz[-2:].tolist())
