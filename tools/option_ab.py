"""GPU tool: same-process A/B of a library test option (default vs the other value) over the layer-object engines at 1080p:
FILM, M2M prepare, IFUNet, IFRNet_L (node default), GMFSS prepare + render.      python tools/option_ab.py <option> <off-value>"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from cfi_amd import _lib as _vfi_lib  # noqa: E402

_vfi_lib.use_test_build()      # the A/B taps live in libvfi_hip_test.so only; one process uses one library, chosen before build() loads it
ge.build()
ge.load_package()
from cfi_amd import _lib, synth  # noqa: E402

lib = _lib.load()
OPT, OFF = sys.argv[1].encode(), int(sys.argv[2])
ON = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H, W = 1080, 1920
fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
x0, x1 = fr[0].cuda().contiguous(), fr[1].cuda().contiguous()
out = torch.empty(H, W, 3, device="cuda")


def timed(fn, n=5):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def ab(name, fn):
    res = {ON: [], OFF: []}
    outs = {}
    for rep in range(3):
        for v in (ON, OFF):
            _lib.check(lib.vfi_test_set_option(OPT, v), "set_option")
            res[v].append(timed(fn))
    _lib.check(lib.vfi_test_set_option(OPT, ON), "set_option")
    print(f"{name:28s} {OPT.decode()}={ON}: {min(res[ON]):8.2f} ms   {OPT.decode()}={OFF}: {min(res[OFF]):8.2f} ms   ({(min(res[ON]) / min(res[OFF]) - 1) * 100:+.1f} %)", flush=True)


from cfi_amd.film import FilmEngine  # noqa: E402

e = FilmEngine(synth.film_synth_state_dict(1234))
ab("FILM forward", lambda: e.forward(x0, x1))
e.close()
from cfi_amd.m2m import M2MEngine  # noqa: E402

e = M2MEngine(synth.m2m_synth_state_dict(1234))
ab("M2M prepare", lambda: e.prepare(x0, x1))
e.close()
from cfi_amd.ifunet import IFUNetEngine  # noqa: E402

e = IFUNetEngine(synth.ifunet_synth_state_dict(1234))
ab("IFUNet forward (ensemble)", lambda: e.forward(x0, x1, 0.5, out, scale=1.0, ensemble=True))
e.close()
from cfi_amd.ifrnet import IFRNetEngine  # noqa: E402

e = IFRNetEngine(synth.ifrnet_synth_state_dict("L", 1234), "L")
o4 = out.view(1, H, W, 3)
ab("IFRNet_L node default", lambda: e.forward([x0], [x1], 0.5, 1.0, o4))
e.close()
from cfi_amd.gmfss import GMFSSEngine  # noqa: E402

e = GMFSSEngine(synth.gmfss_coherent_state_dicts(3, "union"))
tx = synth.texture_frames(2, H, W, seed=5)
g0, g1 = tx[0].cuda().contiguous(), tx[1].cuda().contiguous()
ab("GMFSS prepare", lambda: e.prepare(g0, g1))
ab("GMFSS render", lambda: e.render(0.5, out))
e.close()
