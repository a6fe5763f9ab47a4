"""GPU tool: the RIFE 4.7 step of bench.py (1080p, 32 pairs) with direct-conv tile variants forced by trace name
(include/vfi_hip_test.h: vfi_test_variant_override) — per-kernel HIP-event milliseconds for each override set.
    python tools/rife_variant_ab.py "conv0b_b3=40" "conv0b_b3=40,conv0b_b2=40" ..."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from cfi_amd import _lib, synth  # noqa: E402

_lib.use_test_build()      # the override lives in libvfi_hip_test.so only; one process uses one library, chosen before build() loads it
ge.build()
lib = _lib.load()
from cfi_amd.rife import RifeEngine  # noqa: E402

B, H, W = 32, 1080, 1920
eng = RifeEngine(synth.rife47_synth_state_dict(1234), "4.7")
eng.configure(H, W, B, B + 1, 1.0)
g = torch.Generator(device="cpu").manual_seed(0)
raw = torch.rand((B + 1, H, W, 3), generator=g).cuda()
out = torch.empty((B, H, W, 3), device="cuda")
slot0, slot1, ts = list(range(B)), list(range(1, B + 1)), [0.5] * B
names = ("conv0a_b0", "conv0b_b0", "conv0a_b1", "conv0b_b1", "conv0a_b2", "conv0b_b2", "conv0b_b3", "trans1_conv0a", "encode_batch", "stage_trans4", "stage_trans2", "final_blend")


def step():
    eng.load_frames(list(range(B + 1)), [raw[j] for j in range(B + 1)])
    eng.interpolate(slot0, slot1, ts, out)


ref = None
for spec in [""] + sys.argv[1:]:
    if spec.startswith("opt:"):      # "opt:xcd_bands=0": a library A/B option instead of a variant override
        k, v = spec[4:].split("=")
        assert lib.vfi_test_set_option(k.encode(), int(v)) == 0
    else:
        lib.vfi_test_variant_override(spec.encode())
    try:
        step(); step()
        torch.cuda.synchronize()
        lib.vfi_trace_reset(); lib.vfi_trace_enable(1)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        lib.vfi_trace_enable(0)
        rep = _lib.trace_report()
        lib.vfi_trace_reset()
        tot = sum(v[1] for v in rep.values()) / 3
        o = out[::8, ::64, ::64].clone()
        if ref is None:
            ref = o
        print(f"[{spec or 'default'}] step {tot:.2f} ms  " + "  ".join(f"{k}={rep[k][1] / rep[k][0]:.3f}" for k in names if k in rep) +
              f"  max|d| vs default {float((o - ref).abs().max()):.1e}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"[{spec}] FAILED: {e}", flush=True)
lib.vfi_test_variant_override(b"")
