"""GPU tool: the cycle ledger of the hot Winograd instantiation conv_wino_kernel<8,0,0> (docs/design/winograd.md, "cycle ledger").

Runs the RIFE block-3 ResConv shape (32 x 272x480, 64 -> 64, LeakyReLU: the bench line's dominant kernel) through vfi_conv3x3's
Winograd variant under the test option wino_probe = 0..4 and prints, per wave of workgroup 0, where its shader cycles go:

  probe 1   cycles waiting in sub-step 1's s_waitcnt vmcnt (the wave's OWN activation pieces of chunk k+1)
  probe 2   cycles between "before sub-step 3's s_waitcnt" and "after its s_barrier" (own weight pieces + the workgroup's skew)
  probe 3   per item: set-up (patch / B reads, accumulator clears, first transform), K loop, epilogue
  probe 4   the four sub-steps of a chunk (each issues 16 MFMAs = 1024 matrix-pipe cycles)

Every probe form computes the same output (checked against probe 0) and takes its stamps with s_memtime, consumed at the loop's own
lgkmcnt(0) points; the launch duration of each form is printed next to the unprobed one so the perturbation is on the page.
The shader clock of every launch comes from the clock probe (vfi_clock_probe): cycles and microseconds are both given."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from cfi_amd import _lib as _vfi_lib  # noqa: E402

_vfi_lib.use_test_build()      # the A/B taps live in libvfi_hip_test.so only; one process uses one library, chosen before build() loads it
ge.build()
ge.load_package()
from cfi_amd import _lib  # noqa: E402

lib = _lib.load()
_lib.check(lib.vfi_init(0), "init")
N, H, W, CIN, COUT = 32, 272, 480, 64, 64
if len(sys.argv) > 1 and sys.argv[1].isdigit():
    N = int(sys.argv[1])
MFMA_CYCLES = 64          # v_mfma_f32_32x32x2_f32: 16 passes x 4 cycles
u32 = lambda x: x & 0xFFFFFFFF

g = torch.Generator().manual_seed(5)
x = (torch.rand(N, H, W, CIN, generator=g) - 0.5).cuda()
wt = (torch.rand(COUT, CIN, 3, 3, generator=g) - 0.5) * 0.1
b = torch.rand(COUT, generator=g) - 0.5
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def launch(out):
    _lib.check(lib.vfi_conv3x3(p(x), p(wt), p(b), None, p(out), N, H, W, CIN, COUT, 1, 1, 0.2, 100, None), "vfi_conv3x3")


def run(probe, reps=3):
    _lib.check(lib.vfi_test_set_option(b"wino_probe", probe), "set_option")
    out = torch.empty(N, H, W, COUT, device="cuda")
    launch(out)
    torch.cuda.synchronize()
    rec = torch.zeros((reps, 8), dtype=torch.int64, device="cuda")
    _lib.check(lib.vfi_clock_probe(p(rec), reps), "vfi_clock_probe")
    lib.vfi_trace_reset()
    lib.vfi_trace_enable(1)
    for _ in range(reps):
        launch(out)
    torch.cuda.synchronize()
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    lib.vfi_trace_reset()
    _lib.check(lib.vfi_clock_probe(None, 0), "vfi_clock_probe off")
    sums = (C.c_uint32 * 32)()
    if probe:
        _lib.check(lib.vfi_test_wino_probe_read(sums), "vfi_test_wino_probe_read")
    calls, ms = list(rep.values())[0]
    r = rec.cpu().numpy().astype("uint64")
    cyc = [int(a[2] - a[0]) for a in r]
    ticks = [int(a[3] - a[1]) for a in r]
    _lib.check(lib.vfi_test_set_option(b"wino_probe", 0), "set_option")
    return {"ms": ms / calls, "cycles": cyc, "ticks": ticks, "sums": [list(sums[w * 8:(w + 1) * 8]) for w in range(4)], "out": out}


base = run(0)
mhz = [c / t * 100.0 for c, t in zip(base["cycles"], base["ticks"]) if t]
items_per_wg = None
print(f"shape: {N} x {H}x{W}, {CIN} -> {COUT}, LeakyReLU; Winograd 16x8 regions (variant 100)")
print(f"probe 0 (product kernel): {base['ms'] * 1e3:.1f} us per launch by HIP events; workgroup 0: {base['cycles']} shader cycles, "
      f"{base['ticks']} s_memrealtime ticks -> {', '.join(f'{m:.0f}' for m in mhz)} MHz if the tick is 100 MHz "
      f"(ticks / event time = {sum(base['ticks']) / len(base['ticks']) / (base['ms'] * 1e3):.2f} per us)")
flop_exec = 2.0 * N * H * W * CIN * COUT * 9 / 2.25
print(f"executed MFMA FLOP per launch {flop_exec / 1e9:.2f} G -> {flop_exec / (base['ms'] * 1e-3) / 1e12:.1f} TFLOP/s = {flop_exec / (base['ms'] * 1e-3) / 1e12 / 157.3:.3f} of the 2.4 GHz peak; "
      f"at the measured clock: {flop_exec / (sum(base['cycles']) / len(base['cycles']) * 65536.0):.3f} of 64 FLOP/clk/SIMD x 1024")
res = {}
for probe in (1, 2, 3, 4):
    r = run(probe)
    res[probe] = r
    same = torch.equal(r["out"], base["out"])
    print(f"\nprobe {probe}: {r['ms'] * 1e3:.1f} us per launch ({(r['ms'] / base['ms'] - 1) * 100:+.1f} % vs the product kernel), workgroup 0 {r['cycles'][-1]} cycles; output bit-identical: {same}")
    for w in range(4):
        q0, q1, q2, q3, qn, tend, chunks, pid = r["sums"][w]
        if pid != probe:
            print(f"  wave {w}: no record (probe id {pid})")
            continue
        if probe == 1:
            d = u32(q1 - q0)
            print(f"  wave {w}: {qn} chunks, sub-step 1 vmcnt wait: {d} cycles total = {d / max(qn, 1):.0f} per chunk")
        elif probe == 2:
            d = u32(q1 - q0)
            print(f"  wave {w}: {qn} chunks, sub-step 3 wait + barrier: {d} cycles total = {d / max(qn, 1):.0f} per chunk")
        elif probe == 3:
            su, kl, ep = u32(q1 - q0), u32(q2 - q1), u32(q3 - q2)
            print(f"  wave {w}: {qn} items: set-up {su / max(qn, 1):.0f}, K loop {kl / max(qn, 1):.0f} ({kl / max(chunks, 1):.0f} per chunk; MFMA issue alone {64 * MFMA_CYCLES}), "
                  f"epilogue {ep / max(qn, 1):.0f} cycles per item; sum {(su + kl + ep) / max(qn, 1):.0f} per item, {su + kl + ep} of the workgroup's {r['cycles'][-1]}")
        else:
            s0, s1, s2 = u32(q1 - q0), u32(q2 - q1), u32(q3 - q2)
            print(f"  wave {w}: {qn} chunks: sub-step 0 {s0 / max(qn, 1):.0f}, 1 {s1 / max(qn, 1):.0f}, 2 {s2 / max(qn, 1):.0f} cycles per chunk "
                  f"(16 MFMAs = {16 * MFMA_CYCLES} each); sub-step 3 + the chunk's tail = the K loop per chunk (probe 3) minus these")
print("\nledger (wave 0, per item of 8 chunks; MFMA issue = 512 x 64 = 32768 cycles):")
try:
    q = res[3]["sums"][0]
    n_it = max(q[4], 1)
    su, kl, ep = u32(q[1] - q[0]) / n_it, u32(q[2] - q[1]) / n_it, u32(q[3] - q[2]) / n_it
    v1 = u32(res[1]["sums"][0][1] - res[1]["sums"][0][0]) / max(res[1]["sums"][0][4], 1) * 8
    v2 = u32(res[2]["sums"][0][1] - res[2]["sums"][0][0]) / max(res[2]["sums"][0][4], 1) * 8
    tot = su + kl + ep
    print(f"  item total {tot:.0f} = set-up {su:.0f} + K loop {kl:.0f} + epilogue {ep:.0f}")
    print(f"  K loop {kl:.0f} = MFMA issue 32768 + own-DMA wait {v1:.0f} + weight wait / barrier {v2:.0f} + rest (transform VALU, issue gaps) {kl - 32768 - v1 - v2:.0f}")
    print(f"  shares of the item: MFMA {32768 / tot:.3f}, DMA wait {v1 / tot:.3f}, barrier {v2 / tot:.3f}, K-loop rest {(kl - 32768 - v1 - v2) / tot:.3f}, set-up {su / tot:.3f}, epilogue {ep / tot:.3f}")
    wg = base["cycles"][-1]
    print(f"  workgroup 0 lives {wg} cycles in the product kernel; {n_it} items x {tot:.0f} = {n_it * tot:.0f} ({n_it * tot / wg:.3f}: the probe forms' own cost and the prologue / tail are the difference)")
except Exception as e:  # noqa: BLE001
    print(f"  (incomplete: {type(e).__name__}: {e})")

