"""-m gpu: a poisoned environment (every experiment variable earlier rounds' builds understood — wrong-result ablations included)
does not change one bit of a RIFE 4.7 / M2M result: the library no longer reads them."""
import os
import subprocess
import sys

import pytest
import torch

from test_env_hygiene import POISON

pytestmark = pytest.mark.gpu

_SNIPPET = r"""
import sys, warnings, torch
sys.path.insert(0, {root!r})
from pkgload import load_package
load_package()
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    from cfi_amd import _lib, synth
    lib = _lib.load()
print("WARNED" if any("ignored environment variable" in str(x.message) for x in w) else "QUIET")
from cfi_amd.rife import RifeEngine, run_tasks
from cfi_amd.m2m import M2MEngine
torch.cuda.set_device(0)
outs = []
e = RifeEngine(synth.rife47_synth_state_dict(77), "4.7")
frames = synth.smooth_frames(3, 200, 328, seed=5, shift=3.0)
outs.append(run_tasks(e, frames, [(0, 0.5), (1, 0.3), (0, 0.8)], batch_size=3).cpu())
e.close()
m = M2MEngine(synth.m2m_synth_state_dict(1234))
f = synth.smooth_frames(2, 128, 192, seed=2, shift=4.0).cuda()
m.prepare(f[0], f[1])
outs.append(m.render(0.5).cpu())
m.close()
torch.save(outs, {out!r})
"""


def test_poisoned_environment_changes_nothing(hip_lib, tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res, said = {}, {}
    clean = {k: v for k, v in os.environ.items() if not k.startswith("VFI_")}
    for tag, env in (("clean", clean), ("poisoned", dict(clean, **POISON))):
        out = str(tmp_path / f"{tag}.pt")
        r = subprocess.run([sys.executable, "-c", _SNIPPET.format(root=root, out=out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[tag], said[tag] = torch.load(out), r.stdout
    assert "QUIET" in said["clean"] and "WARNED" in said["poisoned"]
    for a, b in zip(res["clean"], res["poisoned"]):
        assert torch.equal(a, b), float((a - b).abs().max())
