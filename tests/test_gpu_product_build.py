"""-m gpu: the PRODUCT library (libvfi_hip.so — no test taps, what the package / bench.py / smoke() load) against the oracle, in a
child process: the suite itself runs on libvfi_hip_test.so (tests/conftest.py), and one process uses one library.  RIFE 4.7,
FILM and M2M, one small pair each, per-pixel fp32 |d| <= 1e-3 (BASELINE.json north_star)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %r)
from pkgload import load_package
load_package()
import torch
from cfi_amd import _lib, synth
lib = _lib.load()
assert not _lib.is_test_build()
for tap in ("vfi_test_set_option", "vfi_test_variant_override", "vfi_test_conv_algo", "vfi_rife_debug_read", "vfi_m2m_debug_read"):
    assert not hasattr(lib, tap), tap
loaded = [l.split()[-1] for l in open("/proc/self/maps") if "libvfi_hip" in l]
assert loaded and all(p.endswith("libvfi_hip.so") for p in loaded), loaded
torch.cuda.set_device(0)
torch.set_num_threads(min(32, torch.get_num_threads()))
from cfi_amd.rife import RifeEngine, run_tasks
from cfi_amd.film import FilmEngine
from cfi_amd.m2m import M2MEngine
from oracle import rife_oracle, film_oracle, m2m_model_oracle as mo

fr = synth.smooth_frames(2, 180, 320, seed=11, shift=3.0)
sd = synth.rife47_synth_state_dict(1234)
eng = RifeEngine(sd, "4.7")
got = run_tasks(eng, fr, [(0, 0.5)], batch_size=1)
want = rife_oracle.rife_vfi(sd, fr, multiplier=2)[1:2]
e = (got - want).abs().max().item()
print("rife", e); assert e <= 1e-3
eng.close()

x = fr.permute(0, 3, 1, 2).contiguous()
sd = synth.film_synth_state_dict(1234)
eng = FilmEngine(sd)
got = eng.forward(fr[0].cuda().contiguous(), fr[1].cuda().contiguous()).cpu()
with torch.inference_mode():
    want = film_oracle.film_forward(sd, x[0:1], x[1:2])[0].permute(1, 2, 0)
e = (got - want).abs().max().item()
print("film", e); assert e <= 1e-3
eng.close()

sd = synth.m2m_synth_state_dict(1234)
eng = M2MEngine(sd)
got = eng.forward(fr[0].cuda().contiguous(), fr[1].cuda().contiguous(), 0.5).cpu()
with torch.inference_mode():
    want = mo.m2m_forward(sd, x[0:1], x[1:2], [torch.tensor([0.5]).view(1, 1, 1, 1)])[0][0].permute(1, 2, 0)
e = (got - want).abs().max().item()
print("m2m", e); assert e <= 1e-3
eng.close()
print("product-build-ok")
"""


def test_product_library_parity_in_child(hip_lib):
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "product-build-ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
