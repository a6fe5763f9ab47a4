#!/usr/bin/env python3
"""bench.py — interpolated frames/s of the RIFE 4.7 2x hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W        (one process per GPU: the driver's launch)
    python bench.py --gpus N --steps K --warmup W                          (ONE process, N device threads: multidev.py)

One "step" = one pass of the hot path over one batch of a synthetic 1080p frame-pair stream that is
already resident in HBM: for each of B new frames  clamp/pad/encode (vfi_rife_load_frame)  and for
each of the B frame pairs one interpolation at t=0.5 (vfi_rife_interpolate) — B new frames per step (default B = 32: one
pass over the whole 33-frame clip of SURVEY 8d config 2),
outputs written to HBM.  N>1: every rank runs its own stream (weak scaling, pairs are independent)
and the new frames are all-gathered over RCCL/xGMI, overlapped with the next step.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline     — dominant kernel (block3 ResConv 3x3, 64->64 ch @272x480, fp32 MFMA).  The kernel is the Winograd F(2x2,3x3) form
                 (csrc/conv_wino.hip): `achieved` / `frac` count the MFMA FLOP it ISSUES per launch (direct form / 2.25, on whole
                 16x8-pixel regions) / the average launch duration measured with HIP events on the launch stream (library-side
                 tracing, second pass of the same K steps) against the 157.3 TFLOP/s fp32-MFMA peak — a hardware utilisation, <= 1.
                 `algorithmic_equiv` prices the same time in direct-form FLOP (2 * pixels * Cin * Cout * 9, SURVEY 8d: what the
                 layer is worth to any implementation); it exceeds the peak because Winograd skips 5/9 of those multiplications.
  roofline_hbm — the HBM-class kernels (frame pack, transitions with their warps, final blend; M2M's summation splat — inside
                 M2M and on SURVEY 8d config 5's i.i.d. sigma = 8 px field): algorithmic bytes per launch
                 / HIP-event launch duration, against 8.0 TB/s spec and the 6.29 TB/s a float4 copy reaches on this chip.  The
                 M2M cost volume is VALU / LDS bound and is priced against its VALU floor (`bound: "valu"`).
  strong_4k_x4 — BASELINE configs[3]: RIFE 4.9 (arch 4.7), multiplier 4, a 17-frame 2160x3840 host clip = 48 tasks,
                 block-partitioned over the ranks with one halo frame per block (unequal blocks where 48 % N != 0), new frames
                 all-gathered device-side: STRONG scaling (fixed total work), reported beside the weak-scaling headline.
  cpu_baseline — the oracle (torch-CPU restatement, bit-exact vs the reference in the build container) timed on this host's
                 cores BEFORE the GPU leg, on pair 0 of the very clip the GPU leg is timed on (identical tensors, BASELINE.md
                 section 3): a bounded sample (N=1, rank 0 only); host CPU model and core count stated.
  parity       — the in-run gate: frame 0 of the LAST timed step (32-task launch, i.i.d. noise clip) against the oracle's frame
                 for the same pair from the cpu_baseline forwards: max |d|, pixels over 1e-3.  A line whose parity fails is
                 printed with "ok": false and the process exits non-zero.
  clock        — the shader clock the chip sustained inside the dominant kernel during the timed region (s_memtime /
                 s_memrealtime deltas of workgroup 0 in every Winograd launch, vfi_clock_probe) + sysfs sclk / socket power
                 samples where the box exposes them; roofline.frac_at_clock prices the kernel against the peak AT that clock
                 (the chip clocks to its power budget: the same binary measures 5-7 % apart between boxes).
  e2e          — SURVEY 8(d) config 2, PCIe-inclusive (never `value`): a host clip [33,1080,1920,3] fp32
                 (torch.manual_seed(0); torch.rand) through the node class RIFE_VFI.vfi to a host tensor, wall clock from the
                 call to the returned tensor (first H2D ... last D2H + host assembly), warm, median of 3; plus the measured
                 pinned H2D / D2H rates of this box.
  other_paths  — device-resident ms per interpolated 1080p frame of FILM (configs[2]) and M2M (configs[4]) and of the SURVEY 8(f)
                 nodes (GMFSS Fortuna, IFUNet, IFRNet_L) with their direct-form convolution TFLOP/s, same box, same process (not the
                 headline metric).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip table


HBM_SPEC_TBPS = 8.0        # MI355X_MICROARCH.md: HBM3E peak (spec)
HBM_COPY_TBPS = 6.29       # ... measured by a float4 copy on this chip (79 %)


def hbm_entry(kernel, bytes_per_launch, ms_per_launch, launches):
    tbps = bytes_per_launch / (ms_per_launch * 1e-3) / 1e12 if ms_per_launch else float("nan")
    return {"kernel": kernel, "bound": "hbm", "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(ms_per_launch, 4),
            "launches": launches, "achieved": round(tbps, 3), "unit": "TB/s", "peak": HBM_SPEC_TBPS, "frac": round(tbps / HBM_SPEC_TBPS, 4),
            "frac_of_copy_rate": round(tbps / HBM_COPY_TBPS, 4)}


VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9      # 256 CUs x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md): 78.6 T lane-ops/s = 157.3 TFLOP/s / 2


def valu_entry(kernel, lane_ops_per_launch, ms_per_launch, launches, note):
    floor_ms = lane_ops_per_launch / VALU_LANE_OPS_PER_S * 1e3
    return {"kernel": kernel, "bound": "valu", "lane_ops_per_launch": int(lane_ops_per_launch), "avg_launch_ms": round(ms_per_launch, 4),
            "launches": launches, "valu_floor_ms": round(floor_ms, 4), "peak": round(VALU_LANE_OPS_PER_S / 1e12, 1), "unit": "T lane-ops/s",
            "achieved": round(lane_ops_per_launch / (ms_per_launch * 1e-3) / 1e12, 2) if ms_per_launch else None,
            "frac": round(floor_ms / ms_per_launch, 4) if ms_per_launch else None,
            # the same against the UNPACKED VALU rate (16 lanes per clock and SIMD: half of `peak`) — what a kernel whose operations have no
            # packed form can reach (the cost volume's |a - b|: VOP3P carries no abs modifier); `frac` keeps the packed peak
            "frac_unpacked_rate": round(2.0 * floor_ms / ms_per_launch, 4) if ms_per_launch else None, "note": note}


def clip_frames(B):
    """Frames of the resident synthetic clip: SURVEY 8(d) config 2's 33 at the default batch of 32 pairs per step."""
    return B + 1 if B > 16 else 2 * B + 1


def clip_base(B, k):
    """First frame of step parity k: the whole clip every step for B > 16, alternating halves otherwise."""
    return 0 if B > 16 else B * k


def host_cpu_model():
    try:
        import subprocess

        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            if line.startswith("Model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def pcie_rates(dev, nbytes=256 << 20):
    """Pinned H2D / D2H GB/s of this box (one 256 MiB copy each way after a warm-up, HIP events)."""
    h = torch.empty(nbytes // 4, dtype=torch.float32, pin_memory=True)
    d = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out = {}
    for name, (dst, src) in {"h2d": (d, h), "d2h": (h, d)}.items():
        dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src, non_blocking=True)
        e1.record()
        torch.cuda.synchronize(dev)
        out[name] = round(nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9, 2)
    return out


def e2e_leg(sd, dev, H, W, n_frames=33, reps=5, with_u8=True):
    """SURVEY 8(d) config 2 through the drop-in node: host clip in, host tensor out, wall clock."""
    import tempfile

    import cfi_amd.rife as R

    torch.manual_seed(0)
    frames = torch.rand(n_frames, H, W, 3)          # i.i.d. U[0,1): SURVEY's worst-case-gradient clip
    frames8 = (frames * 255).round().to(torch.uint8) if with_u8 else None   # the same clip as 8-bit frames (what video load / save nodes hold)
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "rife47.pth")
        torch.save(sd, pth)
        saved = R.load_file_from_github_release
        R.load_file_from_github_release = lambda model_type, ckpt: pth
        try:
            node = R.RIFE_VFI()
            times, times8, first = [], [], {}
            for clip, acc in ((frames, times), (frames8, times8)) if with_u8 else ((frames, times),):
                for i in range(reps + 2):           # two warm-up calls: checkpoint load, workspace, pinned rings — and the ring / page
                                                    # cache state the second call still settles (it measures 1.5-2x the steady state)
                    t0 = time.perf_counter()
                    with contextlib.redirect_stdout(sys.stderr):       # the node reports on stdout like the reference; stdout is the JSON line's
                        res = node.vfi("rife47.pth", clip, multiplier=2, batch_size=16)
                    dt = time.perf_counter() - t0
                    n_out = res[0].shape[0]
                    del res                         # release of the 1.6 GB result happens outside the timed region
                    if i > 1:
                        acc.append(dt)
                    elif clip is frames:
                        first[i] = dt
        finally:
            R.load_file_from_github_release = saved
            for e in R._model_cache.values():
                e.close()
            R._model_cache.clear()
    med = sorted(times)[len(times) // 2]
    new = n_frames - 1
    if not with_u8:      # the long-clip leg: what the pipeline SUSTAINS once the head / tail of a call are amortised
        return {"workload": f"the same call on a {n_frames}-frame clip -> [{n_out},{H},{W},3]; warm, median of {reps}", "value": round(new / med, 2),
                "unit": "interpolated frames/s (PCIe-inclusive, host tensor to host tensor)", "seconds": [round(t, 4) for t in times],
                "first_call_s": round(first.get(0, float("nan")), 4)}
    med8 = sorted(times8)[len(times8) // 2]
    rates = pcie_rates(dev)
    return {
        "workload": f"RIFE_VFI.vfi('rife47.pth', frames[{n_frames},{H},{W},3] fp32 host, torch.manual_seed(0) torch.rand, multiplier=2) -> host "
                    f"tensor [{n_out},{H},{W},3]; wall clock of the call, warm, median of {reps}",
        "value": round(new / med, 2),
        "unit": "interpolated frames/s (PCIe-inclusive, host tensor to host tensor)",
        "seconds": [round(t, 4) for t in times],
        "spread": round((max(times) - min(times)) / med, 4),
        "first_call_s": round(first.get(0, float("nan")), 4),       # checkpoint load + weight pack + workspace + pinned rings
        "second_call_s": round(first.get(1, float("nan")), 4),      # ring / page-cache state still settling
        "h2d_bytes": n_frames * H * W * 3 * 4,
        "d2h_bytes": new * H * W * 3 * 4,
        "pcie_pinned_GBps": rates,
        "uint8_clip": {
            "note": "same call with the clip as uint8 frames (extension beyond the reference's float32 IMAGE contract, SURVEY 8f rank 1): "
                    "x / 255 and round(y * 255) on the device, uint8 tensor returned, a quarter of the host and PCIe bytes",
            "value": round(new / med8, 2),
            "seconds": [round(t, 4) for t in times8],
        },
    }


@contextlib.contextmanager
def oracle_threads(n=32):
    """Host-side oracle forwards of the extra legs: 32 threads (torch's default of one per logical CPU is several times slower on the
    256-CPU GPU box for these convolution sizes — what cpu_baseline measures)."""
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, min(n, before)))
    try:
        with torch.inference_mode():
            yield
    finally:
        torch.set_num_threads(before)


def leg_parity(got, want_fn, what, keep=None):
    """In-run gate of an extra leg: the frame the leg just produced (host [H,W,3]) against the oracle on the identical host tensors.
    Never raises: an oracle that fails is reported as an error string (the leg's timing stays)."""
    try:
        t0 = time.time()
        with oracle_threads():
            want = want_fn()
        p = parity_of(got, want)
        p["what"] = what
        p["oracle_s"] = round(time.time() - t0, 1)
        if keep is not None:
            keep.append(want)
        return p
    except Exception as e:  # noqa: BLE001
        return {"ok": False, "error": f"{type(e).__name__}: {e}", "what": what}


def pair_lanes_rate(dev, build, step, model, n_pairs=12, first=None):
    """Sustained rate of the node's pair loop with its pair lanes (comfyui-frame-interpolation_amd/lanes.py): K engines on K HIP streams, the
    pairs of a clip round robin over them — what the node classes do for a clip of more than one pair.  step(engine, out_k) issues one
    pair (everything the node runs per pair at multiplier 2) on the current stream.  -> {"lanes", "ms_per_pair", "frames_per_s_2x",
    "vs_one_stream"} (never raises)."""
    try:
        from cfi_amd.lanes import LaneSet, lanes_for

        k = lanes_for(model)
        lanes = LaneSet(build, k, first=first)      # the caller's engine is lane 0 (and stays the caller's)
        try:
            pairs = [lanes.lane(i) for i in range(k)]

            def run(n, width):
                for i in range(n):
                    eng, st = pairs[i % width]
                    with torch.cuda.stream(st):
                        step(eng, i % width)

            rates = {}
            for width in (1, k):
                for e, _ in pairs:      # FILM / IFUNet: an engine's own two-stream fork is for a lone pair (lanes.tell_lone_pair in the node loops)
                    if hasattr(e, "lone_pair"):
                        e.lone_pair(width == 1)
                run(2 * width, width)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                run(n_pairs, width)
                torch.cuda.synchronize(dev)
                rates[width] = (time.perf_counter() - t0) / n_pairs
            return {"lanes": k, "ms_per_pair": round(rates[k] * 1e3, 3), "frames_per_s_2x": round(1 / rates[k], 1),
                    "one_stream_ms_per_pair": round(rates[1] * 1e3, 3), "vs_one_stream": round(rates[1] / rates[k], 3),
                    "what": "%d pairs round robin over %d engines on %d HIP streams (the node loop for clips of > 1 pair; frames bit-identical to one "
                            "stream: tests/test_gpu_pair_lanes.py)" % (n_pairs, k, k)}
        finally:
            if first is not None and hasattr(first, "lone_pair"):
                first.lone_pair(True)
            lanes.close()
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def other_paths(dev, H, W, parity=True):
    """FILM and M2M, device-resident, ms per interpolated frame (BASELINE.json configs[2] / configs[4]); each with a `parity` object:
    the frame of the timed call against the oracle (film_oracle / m2m_model_oracle, bit-exact vs the reference in the build container)
    on the same host tensors, all values, per-pixel |d| <= 1e-3."""
    from cfi_amd import synth
    from cfi_amd.film import FilmEngine
    from cfi_amd.m2m import M2MEngine

    fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
    x0, x1 = fr[0].to(dev).contiguous(), fr[1].to(dev).contiguous()
    xn = fr[..., :3].permute(0, 3, 1, 2).contiguous()      # the oracles' NCHW view of the same pair

    def timed(fn, n):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n

    out = {}
    film_sd = synth.film_synth_state_dict(1234)
    eng = FilmEngine(film_sd)
    t = timed(lambda: eng.forward(x0, x1), 3)
    out["film_2x"] = {"ms_per_frame": round(t * 1e3, 2), "frames_per_s": round(1 / t, 2),
                      "tflops": round(8823.8 * (H * W) / (1080 * 1920) / t / 1e3, 1), "flop_per_frame": "8.82 TFLOP @1080p (SURVEY 8d)"}
    if parity:
        from oracle import film_oracle

        got = eng.forward(x0, x1).cpu()
        out["film_2x"]["parity"] = leg_parity(got, lambda: film_oracle.film_forward(film_sd, xn[0:1], xn[1:2])[0].permute(1, 2, 0),
                                              f"the timed call's frame (smooth pair seed 2, {H}x{W}, t = 0.5) vs oracle.film_oracle.film_forward on the same host tensors")
    out["film_2x"]["pair_lanes"] = pair_lanes_rate(dev, lambda: FilmEngine(film_sd), lambda e, k: e.forward(x0, x1), "film", n_pairs=12, first=eng)
    eng.close()
    m2m_sd = synth.m2m_synth_state_dict(1234)
    eng = M2MEngine(m2m_sd)
    tp = timed(lambda: eng.prepare(x0, x1), 5)
    tr = timed(lambda: eng.render(0.5), 20)
    out["m2m"] = {"prepare_ms_per_pair": round(tp * 1e3, 3), "render_ms_per_frame": round(tr * 1e3, 3),
                  "frames_per_s_2x": round(1 / (tp + tr), 1), "frames_per_s_8x": round(7 / (tp + 7 * tr), 1)}
    m2m_outs = {}

    def m2m_pair(e, k):
        e.prepare(x0, x1)
        m2m_outs[k] = e.render(0.5, m2m_outs.get(k))

    out["m2m"]["pair_lanes"] = pair_lanes_rate(dev, lambda: M2MEngine(m2m_sd), m2m_pair, "m2m", n_pairs=48, first=eng)
    if parity:
        from oracle import m2m_model_oracle as mo

        got = eng.render(0.5).cpu()
        want_keep = []
        out["m2m"]["parity"] = leg_parity(
            got, lambda: mo.m2m_forward(m2m_sd, xn[0:1], xn[1:2], [torch.tensor([0.5]).view(1, 1, 1, 1)])[0][0].permute(1, 2, 0),
            f"the timed prepare + render(0.5) frame (smooth pair seed 2, {H}x{W}) vs oracle.m2m_model_oracle.m2m_forward on the same host tensors", keep=want_keep)
        par = out["m2m"]["parity"]
        if par.get("n_over_1e-3", 0) > 0:
            # M2M's output divides splatted colour by splatted weight: where a pixel's whole weight comes from bilinear factors of ~1e-4 px,
            # a flow that differs in its last bits (another summation order in the convolutions upstream) moves the ORACLE's own frame by more
            # than 1e-3 too.  Such a pixel is judged against the oracle's sensitivity, measured here on this very pair: its frame with the
            # flows entering the splats perturbed by a relative 9e-6 (= this path's measured flow deviation), 3 seeds -> the 99.9 % Poisson
            # quantile of the outlier count (oracle/m2m_hot_certificate.outlier_bound, the rule tests/test_gpu_real_ckpt.py applies)
            try:
                from oracle import m2m_hot_certificate as cert

                t0 = time.time()
                with oracle_threads():
                    _, bound, counts, mean_moved = cert.outlier_bound(m2m_sd, fr, 0.5)
                pix_over = int(((got - want_keep[0]).abs().max(dim=2).values > 1e-3).sum().item())
                par["pixels_over_1e-3"] = pix_over
                par["bound"] = {"pixels": bound, "oracle_outliers_per_seed": counts, "oracle_mean_moved": mean_moved,
                                "rule": "99.9 % Poisson quantile of the ORACLE's own count of pixels moving by > 1e-3 under flows x (1 +- 9e-6), 3 seeds",
                                "certificate": "in-run: oracle/m2m_hot_certificate.outlier_bound on this pair", "oracle_s": round(time.time() - t0, 1)}
                par["ok_plain_gate"] = False
                par["ok"] = bool(pix_over <= bound and par["mean_abs"] <= max(mean_moved, 1e-6))
            except Exception as e:  # noqa: BLE001
                par["bound"] = {"error": f"{type(e).__name__}: {e}"}
        # the one place in the suite where the 1e-3 gate is NOT the criterion, stated here so the line does not hide it
        out["m2m"]["parity"]["hot_checkpoint_exception"] = {
            "applies_to_this_line": False,
            "where": "tests/test_gpu_bocchi.py::test_m2m_full_frame_vs_host_oracle[hot] (synthetic 'hot' checkpoint, refined flows up to 107 px)",
            "n_over_1e-3": 1, "bound": "count <= the smallest count (11) and mean <= the smallest mean of the ORACLE's own frame under flows x (1 +- 9e-6)",
            "certificate": "tests/golden/m2m_hot_certificate.json (oracle/m2m_hot_certificate.py)"}
    # M2M's HBM-class kernels by HIP events (one traced prepare + 4 renders): the render kernel (8 summation splats of [Hp,Wp,4]:
    # input 16 + flow 8 + output 16 B per pixel and splat by SURVEY's definition) and the 9x9 cost volume (per level and direction: two
    # 32-channel feature maps in, 81 channels out)
    from cfi_amd import _lib
    lib = _lib.load()
    lib.vfi_trace_reset()
    lib.vfi_trace_enable(1)
    eng.prepare(x0, x1)
    for _ in range(4):
        eng.render(0.5)
    torch.cuda.synchronize(dev)
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    lib.vfi_trace_reset()
    hp, wp = -(-H // 64) * 64, -(-W // 64) * 64
    hbm = []
    if "m2m_render" in rep:
        calls, ms = rep["m2m_render"]
        # SURVEY 8(d)'s unit for this kernel class: a summation splat moves (4 in + 2 flow + 4 out) x 4 B per pixel, M2M renders 8 of them
        # per frame = 668.5 MB at 1080p.  Round 6: ONE kernel does the 8 splats AND what surrounded them (the splat inputs and
        # forwarp_mframe_mask's combine), so the bytes it has to move are fewer than that definition: tf 8 B + e 4 B per splat and pixel,
        # the normalised image 16 B per direction and pixel, the RGB frame out — `fused_compulsory_bytes`; `frac` stays on SURVEY's figure.
        e = hbm_entry("m2m_render (M2M render as one kernel: splat inputs + 8 summation splats [%d,%d,4] + combine; csrc/m2m_render.hip); bytes = SURVEY "
                      "8(d)'s 8 x 40 B per pixel" % (hp, wp), 8 * 40.0 * hp * wp, ms / calls, calls)
        e["fused_compulsory_bytes"] = int((8 * 12 + 2 * 16) * hp * wp + 12 * H * W)
        try:      # HBM bytes from the committed PMC passes of this kernel (they cannot share a run with this timing), scaled to this size
            tj = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))["m2m_render"]
            e["traffic"] = int(tj["bytes_per_launch"] * (hp * wp) / float(tj["pixels"]))
            e["traffic_over_algorithmic"] = round(e["traffic"] / e["algorithmic_bytes_per_launch"], 3)
            e["traffic_over_fused_compulsory"] = round(e["traffic"] / e["fused_compulsory_bytes"], 3)
            e["traffic_source"] = "profiles/roofline_traffic.json (" + tj["source"] + "), not measured in this run"
        except Exception:  # noqa: BLE001
            e["traffic"] = None
        hbm.append(e)
    if "costvol9x9" in rep:
        calls, ms = rep["costvol9x9"]
        levels = [(hp >> k, wp >> k) for k in range(2, 7)]             # 272x480 ... 17x30, both directions in one launch
        tot = sum(2 * (2 * 32 * 4 + 81 * 4) * h * w for h, w in levels)
        # 81 displacements x 32 channels x (v_sub_f32 + v_add_f32 |x|) per pixel and direction; the LDS floor (81 x 8 ds_read_b128 per
        # pixel at 4 LDS cycles per wave instruction) is the same 23 us — an HBM fraction could never approach 1 for this kernel
        ops = sum(2 * 81 * 32 * 2 * h * w for h, w in levels)
        e = valu_entry("costvol9x9 (M2M prepare: 5 pyramid levels %s, 32 channels, both directions per launch; work and time summed over "
                       "the levels)" % "/".join(f"{h}x{w}" for h, w in levels), ops, ms / calls * 5, calls // 5,
                       "2 VALU lane-ops per (displacement, channel); algorithmic HBM bytes %d (%.3f TB/s): not the bound; the coarse levels "
                       "(4 .. 60 workgroups) are launch-latency bound" % (tot, tot / (ms / calls * 5 * 1e-3) / 1e12))
        hbm.append(e)
    eng.close()
    # the splat micro-benchmark — in [1,4,1088,1920] U[0,1), flow i.i.d. N(0, 8 px) seed 2 (an INCOHERENT field:
    # the stress case of cupy_ops/softsplat.py:140-192), 8 launches; all passes of a launch are summed
    try:
        import ctypes as C_

        g = torch.Generator(device="cpu").manual_seed(2)
        x = torch.rand(1, hp, wp, 4, generator=g).to(dev)
        fl = (torch.randn(1, hp, wp, 2, generator=g) * 8.0).to(dev)
        o = torch.empty_like(x)
        pp = lambda t: C_.c_void_p(t.data_ptr())
        _lib.check(lib.vfi_softsplat_sum(pp(x), pp(fl), pp(o), 1, hp, wp, 4, _lib.stream_ptr()), "vfi_softsplat_sum")
        torch.cuda.synchronize(dev)
        lib.vfi_trace_reset()
        lib.vfi_trace_enable(1)
        for _ in range(8):
            _lib.check(lib.vfi_softsplat_sum(pp(x), pp(fl), pp(o), 1, hp, wp, 4, _lib.stream_ptr()), "vfi_softsplat_sum")
        torch.cuda.synchronize(dev)
        lib.vfi_trace_enable(0)
        rep = _lib.trace_report()
        lib.vfi_trace_reset()
        calls = rep["softsplat_sum"][0]
        ms = sum(v[1] for v in rep.values())
        e = hbm_entry("softsplat_sum micro-benchmark (SURVEY 8d config 5: [1,%d,%d,4], flow i.i.d. N(0, 8 px) seed 2; every pass of a launch: %s)"
                      % (hp, wp, ", ".join(f"{k} {v[1] / v[0] * 1e3:.1f} us" for k, v in rep.items())), (4 + 2 + 4) * 4.0 * hp * wp, ms / calls, calls)
        hbm.append(e)
    except Exception as ex:      # never lose the line to an extra leg
        hbm.append({"kernel": "softsplat_sum micro-benchmark", "error": f"{type(ex).__name__}: {ex}"})
    out["roofline_hbm"] = hbm
    return out


def other_nodes(dev, H, W, parity=True):
    """The SURVEY 8(f) nodes — GMFSS Fortuna (union), IFUNet, IFRNet_L — device-resident at the bench resolution: ms per interpolated
    frame and the direct-form convolution TFLOP/s they sustain (same box, same process; not the headline metric).  Each leg is
    independent: one that fails is reported as an error string."""
    from cfi_amd import synth

    fr = synth.smooth_frames(2, H, W, seed=2, shift=4.0)
    x0, x1 = fr[0].to(dev).contiguous(), fr[1].to(dev).contiguous()
    out = torch.empty(H, W, 3, device=dev)
    res = {}

    def timed(fn, n, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n

    def counted(eng, fn):
        eng.conv_flop = 0.0
        fn()
        f, eng.conv_flop = eng.conv_flop, None
        return f

    try:      # GMFSS Fortuna: matched ("coherent") weights + a textured pair, so that GMFlow finds a true small motion as a trained model does
        from cfi_amd.gmfss import GMFSSEngine

        # (checkpoint seed 1234 + texture cell 16 seed 2: the vector of tests/test_gpu_gmfss.py::test_end_to_end_gate_1080p, on which the
        # ORACLE's own output moves by < 2e-4 under rounding-level input noise — GMFlow's softmax matching makes other vectors flip matches)
        gm_sds = synth.gmfss_coherent_state_dicts(1234, "union")
        eng = GMFSSEngine(gm_sds)
        tx = synth.texture_frames(4, H, W, seed=2, cell=16)[:2].contiguous()
        g0, g1 = tx[0].to(dev).contiguous(), tx[1].to(dev).contiguous()
        tp = timed(lambda: eng.prepare(g0, g1), 3)
        tr = timed(lambda: eng.render(0.5, out), 5)
        fp, frn = counted(eng, lambda: eng.prepare(g0, g1)), counted(eng, lambda: eng.render(0.5, out))
        res["gmfss_fortuna_union"] = {"prepare_ms_per_pair": round(tp * 1e3, 2), "render_ms_per_frame": round(tr * 1e3, 2),
                                      "frames_per_s_2x": round(1 / (tp + tr), 1), "conv_gflop_direct_form": {"prepare": round(fp / 1e9, 1), "render": round(frn / 1e9, 1)},
                                      "conv_tflops_direct_form": round((fp + frn) / (tp + tr) / 1e12, 1),
                                      "note": "convolution FLOP only (GMFlow's attention matmuls and the splats are not counted)"}
        gm_outs = {}
        if parity:
            from oracle import gmfss_oracle

            eng.prepare(g0, g1)
            eng.render(0.5, out)
            gx = tx.permute(0, 3, 1, 2).contiguous()
            res["gmfss_fortuna_union"]["parity"] = leg_parity(
                out.cpu(), lambda: gmfss_oracle.gmfss_forward(gm_sds, gx[0:1], gx[1:2], 0.5).permute(0, 2, 3, 1)[0],
                f"the timed prepare + render(0.5) frame (coherent checkpoint 1234, texture pair cell 16 seed 2, {H}x{W}) vs oracle.gmfss_oracle.gmfss_forward on the same host tensors")

        def gm_pair(e, k):
            if k not in gm_outs:
                gm_outs[k] = torch.empty(H, W, 3, device=dev)
            e.prepare(g0, g1)
            e.render(0.5, gm_outs[k])

        res["gmfss_fortuna_union"]["pair_lanes"] = pair_lanes_rate(dev, lambda: GMFSSEngine(gm_sds), gm_pair, "gmfss", n_pairs=18, first=eng)
        eng.close()
        del eng
    except Exception as e:  # noqa: BLE001
        res["gmfss_fortuna_union"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        from cfi_amd.ifunet import IFUNetEngine

        eng = IFUNetEngine(synth.ifunet_synth_state_dict(1234))
        t = timed(lambda: eng.forward(x0, x1, 0.5, out, scale=1.0, ensemble=True), 3)
        f = counted(eng, lambda: eng.forward(x0, x1, 0.5, out, scale=1.0, ensemble=True))
        res["ifunet"] = {"ms_per_frame": round(t * 1e3, 2), "frames_per_s": round(1 / t, 1), "ensemble": True, "conv_gflop_direct_form": round(f / 1e9, 1),
                         "conv_tflops_direct_form": round(f / t / 1e12, 1)}
        iu_sd, iu_outs = synth.ifunet_synth_state_dict(1234), {}
        if parity:
            from oracle import ifunet_oracle

            eng.forward(x0, x1, 0.5, out, scale=1.0, ensemble=True)
            ix = fr.permute(0, 3, 1, 2).contiguous()
            res["ifunet"]["parity"] = leg_parity(
                out.cpu(), lambda: ifunet_oracle.ifunet_forward(iu_sd, ix[0:1], ix[1:2], 0.5, 1.0, True).permute(0, 2, 3, 1)[0],
                f"the timed call's frame (smooth pair seed 2, {H}x{W}, t = 0.5, ensemble) vs oracle.ifunet_oracle.ifunet_forward on the same host tensors")

        def iu_pair(e, k):
            if k not in iu_outs:
                iu_outs[k] = torch.empty(H, W, 3, device=dev)
            e.forward(x0, x1, 0.5, iu_outs[k], scale=1.0, ensemble=True)

        res["ifunet"]["pair_lanes"] = pair_lanes_rate(dev, lambda: IFUNetEngine(iu_sd), iu_pair, "ifunet", n_pairs=18, first=eng)
        eng.close()
        del eng
    except Exception as e:  # noqa: BLE001
        res["ifunet"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        from cfi_amd.ifrnet import IFRNetEngine

        eng = IFRNetEngine(synth.ifrnet_synth_state_dict("L", 1234), "L")
        o4 = out.view(1, H, W, 3)
        t = timed(lambda: eng.forward([x0], [x1], 0.5, 1.0, o4), 5)      # the node's default call (multiplier 2): working resolution 0.5, embedding 1.0
        res["ifrnet_L"] = {"ms_per_frame": round(t * 1e3, 2), "frames_per_s": round(1 / t, 1), "call": "node default (multiplier 2): working resolution x0.5",
                           "conv_tflops_direct_form": round(0.80 * (H * W) / (1080 * 1920) / t, 1), "flop_per_frame": "0.80 TFLOP @1080p (docs/design/ifrnet.md)"}
        ir_sd, ir_outs = synth.ifrnet_synth_state_dict("L", 1234), {}
        if parity:
            from oracle import ifrnet_oracle

            eng.forward([x0], [x1], 0.5, 1.0, o4)
            rx = fr.permute(0, 3, 1, 2).contiguous()
            res["ifrnet_L"]["parity"] = leg_parity(
                o4[0].cpu(), lambda: ifrnet_oracle.ifrnet_forward(ir_sd, rx[0:1], rx[1:2], 0.5, 1.0).permute(0, 2, 3, 1)[0],
                f"the timed call's frame (smooth pair seed 2, {H}x{W}, node default: working resolution x0.5) vs oracle.ifrnet_oracle.ifrnet_forward on the same host tensors")

        def ir_pair(e, k):
            if k not in ir_outs:
                ir_outs[k] = torch.empty(1, H, W, 3, device=dev)
            e.forward([x0], [x1], 0.5, 1.0, ir_outs[k])

        res["ifrnet_L"]["pair_lanes"] = pair_lanes_rate(dev, lambda: IFRNetEngine(ir_sd, "L"), ir_pair, "ifrnet", n_pairs=36, first=eng)
        eng.close()
        del eng
    except Exception as e:  # noqa: BLE001
        res["ifrnet_L"] = {"error": f"{type(e).__name__}: {e}"}
    torch.cuda.empty_cache()
    return res


def other_paths_dist(dev, H, W, world, rank, backend):
    """N > 1 (one process per GPU): FILM 2x and M2M 2x with the frame PAIRS sharded over the ranks (each rank interpolates its own
    pair stream — weak scaling, like the headline metric — and the new frames are all-gathered over RCCL), timed
    between barriers, max over ranks.  Gives the driver's multi-GPU run the FILM / M2M curves next to RIFE's."""
    import torch.distributed as dist

    from cfi_amd import synth
    from cfi_amd.film import FilmEngine
    from cfi_amd.m2m import M2MEngine

    fr = synth.smooth_frames(2, H, W, seed=2 + rank, shift=4.0)
    x0, x1 = fr[0].to(dev).contiguous(), fr[1].to(dev).contiguous()
    gathered = torch.empty((world, H, W, 3), dtype=torch.float32, device=dev) if backend == "nccl" else None

    def timed(fn, n):
        fn()
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            out = fn()
            if gathered is not None:
                dist.all_gather_into_tensor(gathered, out.view(1, H, W, 3))
        torch.cuda.synchronize(dev)
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / n

    out = {}
    eng = FilmEngine(synth.film_synth_state_dict(1234))
    t = timed(lambda: eng.forward(x0, x1), 3)
    out["film_2x"] = {"frames_per_s": round(world / t, 2), "ms_per_pair_per_gpu": round(t * 1e3, 2), "sharding": "frame pairs over ranks, all-gather of new frames"}
    eng.close()
    eng = M2MEngine(synth.m2m_synth_state_dict(1234))

    def m2m_pair():
        eng.prepare(x0, x1)
        return eng.render(0.5)

    t = timed(m2m_pair, 5)
    out["m2m_2x"] = {"frames_per_s": round(world / t, 1), "ms_per_pair_per_gpu": round(t * 1e3, 3), "sharding": "frame pairs over ranks, all-gather of new frames"}
    eng.close()
    return out


def peer_copy_gather_leg(eng, raw, B, H, W, K, dev, world, rank, backend):
    """The headline loop once more with the new frames exchanged WITHOUT a collective kernel: every rank maps its peers' gather buffers
    through IPC handles and pushes its B new frames into them with plain device-to-device copies on a side stream (peer copies over
    xGMI run on the SDMA engines: no compute units taken from the persistent kernels — the form DESIGN.md section 6 prefers on
    point-to-point links).  Reported NEXT to the RCCL-based `value`, never instead of it: this leg has only ever run with its ranks
    on one GPU (tests/test_gpu_bench_line.py), so it sits behind the watchdog and validates what arrived before it reports a number."""
    import torch.distributed as dist

    ctl = dev if backend == "nccl" else "cpu"
    outs = [torch.empty((B, H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.zeros((world * B, H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    torch.cuda.synchronize(dev)
    views, err = [], None          # views[r][k]: rank r's gathered[k] as a tensor of THIS process
    try:
        mine = [g.untyped_storage()._share_cuda_() for g in gathered]
    except Exception as e:
        mine, err = None, f"{type(e).__name__}: {e}"
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if err is None and all(h is not None for h in everyone):
        try:
            for r in range(world):
                if r == rank:
                    views.append(gathered)
                    continue
                vk = []
                for k in range(2):
                    st = torch.UntypedStorage._new_shared_cuda(*everyone[r][k])
                    vk.append(torch.empty(0, dtype=torch.float32, device=st.device).set_(st, 0, (world * B, H, W, 3)))
                views.append(vk)
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
    elif err is None:
        err = "a peer could not export its buffers"
    bad = torch.tensor([0.0 if err is None else 1.0], device=ctl)
    dist.all_reduce(bad)            # every rank learns whether ALL mappings exist before any of them enters the loop's barriers
    if float(bad.item()) != 0.0:
        return {"error": err or "a peer could not map the buffers"}
    side = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    copied = [None, None]           # outs[k]'s copies of two steps ago have left it
    slot0, slot1, ts = list(range(B)), list(range(1, B + 1)), [0.5] * B

    def step(i):
        k = i & 1
        base = clip_base(B, k)
        if copied[k] is not None:
            main.wait_event(copied[k])
        eng.load_frames(list(range(B + 1)), [raw[base + j] for j in range(B + 1)])
        eng.interpolate(slot0, slot1, ts, outs[k])
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for r in range(world):          # own block first, then the peers, each rank starting with a different one
                views[(rank + r) % world][k][rank * B:(rank + 1) * B].copy_(outs[k], non_blocking=True)
            copied[k] = torch.cuda.Event()
            copied[k].record(side)

    def timed(n):
        torch.cuda.synchronize(dev)
        dist.barrier()
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        torch.cuda.synchronize(dev)
        dist.barrier()
        return time.perf_counter() - t0

    timed(2)
    # validation: every rank publishes a fingerprint of its own new frames; what arrived in THIS rank's buffer must match all of them
    fp = [float(outs[1][0, ::97, ::89].double().sum().item()), float(outs[1][B - 1, ::83, ::101].double().sum().item())]
    fps = [None] * world
    dist.all_gather_object(fps, fp)
    ok = all(abs(float(gathered[1][r * B, ::97, ::89].double().sum().item()) - fps[r][0]) <= 1e-6 * max(1.0, abs(fps[r][0])) and
             abs(float(gathered[1][r * B + B - 1, ::83, ::101].double().sum().item()) - fps[r][1]) <= 1e-6 * max(1.0, abs(fps[r][1]))
             for r in range(world))
    flag = torch.tensor([0.0 if ok else 1.0], device=ctl)
    dist.all_reduce(flag)
    if float(flag.item()) != 0.0:
        return {"error": f"{int(flag.item())} rank(s) did not find their peers' frames in the mapped buffers"}
    t = torch.tensor([timed(K)], dtype=torch.float64, device=ctl)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    del views
    return {"value": round(world * B * K / el, 2), "unit": "interpolated frames/s (whole job)", "ms_per_step": round(el / K * 1e3, 3), "steps": K,
            "exchange": "each rank copies its new frames into every peer's gather buffer (IPC-mapped) with device-to-device copies on a side stream; "
                        "no collective kernel", "validated": "fingerprints of every rank's frames found in every rank's buffer"}


def strong_4k_x4(args, dev, world, rank, backend, group=None):
    """BASELINE.json configs[3] / SURVEY 8(d) config 4: RIFE 4.9 (= arch 4.7, rife/__init__.py CKPT_NAME_VER_DICT), multiplier 4, a
    17-frame 2160x3840 host clip -> 16 pairs x 3 timesteps = 48 independent tasks (rife/__init__.py:164-174).  STRONG scaling: the
    task list is block-partitioned over the ranks (schedule.shard_tasks: 48 = 8 x 6, but 7 + 7 + ... at N = 5, 6, 7 and 24 / 12 at
    N = 2 / 4), every rank uploads only the frames its block touches (its pairs + one halo frame), interpolates with the node's own
    host pipeline (rife.run_tasks: pinned staging, uploads ahead of compute) and the new frames are all-gathered device-side over
    RCCL.  Timed between barriers from host clip to gathered device frames, max over ranks; PCIe-inclusive on the input side.
    ``group`` (one process, N device threads): the node's multidev path instead — every device copies its shard into the shared
    host output tensor (RifeDeviceGroup.run -> run_sharded)."""
    from cfi_amd import synth
    from cfi_amd.dist import all_gather_frames
    from cfi_amd.rife import RifeEngine, effective_batch, run_tasks
    from cfi_amd.schedule import rife_task_list, shard_tasks

    H, W, n_frames, mult = args.strong_height, args.strong_width, args.strong_frames, 4
    g = torch.Generator(device="cpu").manual_seed(0)
    frames = torch.rand((n_frames, H, W, 3), generator=g, dtype=torch.float32)      # the same clip on every rank
    _, tasks = rife_task_list(n_frames, mult, None)
    bounds = [shard_tasks(tasks, r, world) for r in range(world)]
    counts = [hi - lo for lo, hi in bounds]
    lo, hi = bounds[rank]
    bs = effective_batch(1, H, W, max(counts))
    kept = [None]              # the gathered frames of the last repetition (parity gate below)
    sd49 = synth.rife47_synth_state_dict(49)
    if group is not None:      # single process, device threads
        from cfi_amd.hostpipe import prefault_async

        out = torch.empty((len(tasks), H, W, 3), dtype=torch.float32)
        for f in prefault_async(out):
            f.result()

        def once():
            group.run(frames, tasks, bs, 1.0, out, list(range(len(tasks))))
    else:
        import torch.distributed as dist

        def agree(err, where):
            """A rank that fails before a collective would leave the others waiting in it for ever: every local step runs under
            try, the ranks agree on its outcome (one small all-reduce), and only then go on — or ALL raise."""
            if world > 1:
                flag = torch.tensor([0.0 if err is None else 1.0], device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if flag.item() > 0:
                    raise RuntimeError(f"strong_4k_x4: a rank failed in {where} ({type(err).__name__ if err else 'another rank'}: {err})")
            elif err is not None:
                raise err

        eng, err0 = None, None
        try:
            eng = RifeEngine(sd49, "4.7", device=dev)
        except Exception as e:  # noqa: BLE001
            err0 = e
        agree(err0, "engine creation")

        def once():
            err, local = None, None
            try:
                local = run_tasks(eng, frames, tasks[lo:hi], bs, 1.0, out_device=True)
                torch.cuda.synchronize(dev)
            except Exception as e:  # noqa: BLE001
                err = e
            agree(err, "its block of the task list")
            gathered = all_gather_frames(local, counts) if world > 1 else local
            torch.cuda.synchronize(dev)
            kept[0] = gathered
            return gathered.shape[0]

    def sync():
        if group is None and world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize(dev)

    once()                     # warm-up: workspace for 4K, pinned rings
    times = []
    for _ in range(args.strong_reps):
        sync()
        t0 = time.perf_counter()
        once()
        sync()
        dt = time.perf_counter() - t0
        if group is None and world > 1:
            import torch.distributed as dist

            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        times.append(dt)
    if group is None:
        eng.close()
    best = sorted(times)[len(times) // 2]      # the median (upper of two), like the e2e leg; all times are listed under `seconds`
    par = None
    if rank == 0 and kept[0] is not None and not getattr(args, "no_parity", False):
        # in-run gate of this leg: the first and the last gathered frame of the last repetition (with N ranks the last one was computed
        # by rank N-1 and came through the all-gather) vs the oracle on the same host frames
        from oracle import rife_oracle

        ps = []
        for ti in sorted({0, len(tasks) - 1}):
            pair, tt = tasks[ti]
            got = kept[0][ti].cpu()
            x0 = frames[pair].permute(2, 0, 1).unsqueeze(0).contiguous()
            x1 = frames[pair + 1].permute(2, 0, 1).unsqueeze(0).contiguous()
            ps.append(leg_parity(got, lambda: rife_oracle.ifnet47_forward(sd49, x0, x1, torch.tensor([float(tt)]).view(1, 1, 1, 1)).clamp(0, 1)[0].permute(1, 2, 0),
                                 f"task {ti} (pair {pair}, t = {tt:.4g})"))
        if all("error" not in p_ for p_ in ps):
            par = {"max_abs": max(p_["max_abs"] for p_ in ps), "n_over_1e-3": sum(p_["n_over_1e-3"] for p_ in ps), "values": sum(p_["values"] for p_ in ps),
                   "tol": 1e-3, "ok": all(p_["ok"] for p_ in ps), "oracle_s": sum(p_["oracle_s"] for p_ in ps),
                   "what": "gathered frames of " + " and ".join(p_["what"] for p_ in ps) + f" of the last repetition vs oracle.rife_oracle.ifnet47_forward on the same {H}x{W} host frames"}
        else:
            par = next(p_ for p_ in ps if "error" in p_)
    kept[0] = None
    return {
        "parity": par,
        "workload": f"RIFE 4.9 (arch 4.7) x{mult}, {n_frames}-frame {H}x{W} host clip (torch.manual_seed(0) torch.rand) = {len(tasks)} tasks; "
                    f"contiguous task blocks per rank {counts} (+ 1 halo frame each), {bs} tasks per launch",
        "scaling": "strong",
        "value": round(len(tasks) / best, 2),
        "value_is": f"tasks / median of {len(times)} timed repetitions",
        "unit": "interpolated frames/s (whole job; host clip in, " + ("host tensor out" if group is not None else "gathered device frames out") + ")",
        "n_gpus": world,
        "seconds": [round(t, 4) for t in times],
        "tasks_per_rank": counts,
        "collective": "none (N = 1)" if world == 1 else ("each device copies its own shard into the shared host output (multidev.run_sharded)" if group is not None
                       else ("all_gather_into_tensor over RCCL, blocks padded to the largest" if backend == "nccl" else "all_gather (gloo, host)")),
    }



class ClockProbe:
    """vfi_clock_probe wrapper: one record per Winograd launch while installed (include/vfi_hip.h)."""

    def __init__(self, dev, capacity):
        import ctypes as C_

        from cfi_amd import _lib

        self.lib, self._lib = _lib.load(), _lib
        self.rec = torch.zeros((capacity, 8), dtype=torch.int64, device=dev)
        _lib.check(self.lib.vfi_clock_probe(C_.c_void_p(self.rec.data_ptr()), capacity), "vfi_clock_probe")

    def finish(self):
        """-> {trace name: [(cycles, ticks), ...]} of the launches that completed a record."""
        torch.cuda.synchronize()
        self._lib.check(self.lib.vfi_clock_probe(None, 0), "vfi_clock_probe")
        names = self._lib.clock_probe_names()
        r = self.rec.cpu().numpy().astype("uint64")
        out = {}
        for row in r:
            t0, r0, t1, r1, tag = (int(v) for v in row[:5])
            if t1 > t0 and r1 > r0 and tag < len(names):
                out.setdefault(names[tag], []).append((t1 - t0, r1 - r0))
        return out


def clock_summary(by_name, dom):
    def mhz(pairs):
        return [c / t * 100.0 for c, t in pairs if t > 0]

    d = mhz(by_name.get(dom, []))
    allw = [m for v in by_name.values() for m in mhz(v)]
    if not d:
        return None
    return {"shader_mhz": round(sum(d) / len(d), 1), "min": round(min(d), 1), "max": round(max(d), 1), "launches": len(d),
            "all_winograd_launches_mhz": round(sum(allw) / len(allw), 1) if allw else None,
            "avg_ticks_per_launch": round(sum(t for _, t in by_name[dom]) / len(by_name[dom]), 1),
            "avg_cycles_per_launch": round(sum(c for c, _ in by_name[dom]) / len(by_name[dom]), 1)}


class SysfsSampler:
    """Side thread: sclk (MHz) and socket power (W) of the GPU from sysfs every ~10 ms while running.  Best effort — a box that does
    not expose the files yields None; the in-kernel clock above does not depend on it."""

    def __init__(self, dev_index=0):
        import glob
        import threading

        self.freq_files, self.power_files, self.dpm_files = [], [], []
        try:
            cards = []
            for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
                try:
                    if open(os.path.join(c, "device/vendor")).read().strip() == "0x1002":
                        cards.append(c)
                except OSError:
                    pass
            if cards:
                c = cards[min(dev_index, len(cards) - 1)]
                try:      # the card whose PCI address is this process's device (a box exposes all its GPUs in sysfs, visible to HIP or not)
                    pr = torch.cuda.get_device_properties(dev_index)
                    want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
                    for cand in cards:
                        if os.path.basename(os.path.realpath(os.path.join(cand, "device"))).startswith(want):
                            c = cand
                            break
                except Exception:
                    pass
                self.freq_files = sorted(glob.glob(os.path.join(c, "device/hwmon/hwmon*/freq1_input")))
                self.power_files = sorted(glob.glob(os.path.join(c, "device/hwmon/hwmon*/power1_average")) +
                                          glob.glob(os.path.join(c, "device/hwmon/hwmon*/power1_input")))
                self.dpm_files = [f for f in [os.path.join(c, "device/pp_dpm_sclk")] if os.path.exists(f)]
        except Exception:
            pass
        self.sclk, self.power = [], []
        self._stop = threading.Event()
        self._thr = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                if self.freq_files:
                    self.sclk.append(int(open(self.freq_files[0]).read()) / 1e6)
                elif self.dpm_files:
                    for line in open(self.dpm_files[0]):
                        if "*" in line:
                            self.sclk.append(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
                if self.power_files:
                    self.power.append(int(open(self.power_files[0]).read()) / 1e6)
            except Exception:
                pass
            self._stop.wait(0.01)

    def __enter__(self):
        if self.freq_files or self.dpm_files or self.power_files:
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        if self._thr.is_alive():
            self._thr.join(timeout=1.0)

    def summary(self):
        avg = lambda v: round(sum(v) / len(v), 1) if v else None
        return {"sclk_mhz": avg(self.sclk), "sclk_mhz_min": round(min(self.sclk), 1) if self.sclk else None, "socket_power_w": avg(self.power),
                "samples": max(len(self.sclk), len(self.power)),
                "source": (self.freq_files or self.dpm_files or ["-"])[0] + " | " + (self.power_files or ["-"])[0]}


def parity_of(got, want, tol=1e-3):
    """got / want: [H,W,3] fp32 (host).  The north-star gate: per-pixel |d| <= 1e-3."""
    d = (got - want).abs()
    return {"max_abs": float(d.max().item()), "mean_abs": float(d.mean().item()), "n_over_1e-3": int((d > tol).sum().item()),
            "values": int(d.numel()), "tol": tol, "ok": bool((d <= tol).all().item())}

def cpu_baseline(sd, pairs, budget_s=25.0, timing=True):
    """Oracle on the host cores, on the pairs [(f0, f1), ...] the GPU leg's parity check looks at ([H,W,3] fp32 host tensors, t = 0.5).
    Timing on pairs[0]: for a few thread counts (all cores is often NOT the fastest on a many-core host), 1 warm-up + 3 timed 1080p
    forwards each; reports the best median.  Returns (dict, [the oracle's frame [H,W,3] per pair, clamped as the node clamps it]); the
    other pairs take one forward each at the best thread count.  ``timing`` False: one forward per pair, no baseline figure (N > 1
    runs: the parity reference only)."""
    from oracle import rife_oracle

    f0, f1 = pairs[0]
    H, W = f0.shape[0], f0.shape[1]
    x0 = f0.permute(2, 0, 1).unsqueeze(0).contiguous()
    x1 = f1.permute(2, 0, 1).unsqueeze(0).contiguous()
    ts = torch.tensor([0.5]).view(1, 1, 1, 1)
    default = torch.get_num_threads()
    cands = sorted({default, max(1, default // 2), max(1, default // 4)}, reverse=True) if timing else [default]
    best = None
    tried = []
    ref = None
    t_begin = time.time()
    with torch.inference_mode():
        for nt in cands:
            if best is not None and time.time() - t_begin > budget_s:
                break
            torch.set_num_threads(nt)
            times = []
            for i in range(4 if timing else 1):
                t0 = time.time()
                o = rife_oracle.ifnet47_forward(sd, x0, x1, ts)
                if i > 0 or not timing:
                    times.append(time.time() - t0)
                if ref is None:
                    ref = o.clamp(0, 1)[0].permute(1, 2, 0).contiguous()      # rife/__init__.py: the node clamps the new frame
            med = sorted(times)[len(times) // 2]
            tried.append((nt, round(med, 3)))
            if best is None or med < best[1]:
                best = (nt, med)
    refs = [ref]
    torch.set_num_threads(best[0])
    with torch.inference_mode():
        for a, b in pairs[1:]:
            o = rife_oracle.ifnet47_forward(sd, a.permute(2, 0, 1).unsqueeze(0).contiguous(), b.permute(2, 0, 1).unsqueeze(0).contiguous(), ts)
            refs.append(o.clamp(0, 1)[0].permute(1, 2, 0).contiguous())
    torch.set_num_threads(default)
    if not timing:
        return None, refs
    return {
        "value": round(1.0 / best[1], 4),
        "unit": "interpolated frames/s",
        "cores": best[0],
        # BASELINE.md section 3 says "torch.set_num_threads(os.cpu_count())"; this is a STRONGER baseline than that: the best of a few
        # thread counts (all logical CPUs of a many-core host is several times slower for these convolution sizes)
        "cores_policy": f"best of {cands} threads by median (deviates from BASELINE.md section 3's os.cpu_count() = {os.cpu_count()}: that setting is in `tried` "
                        f"when torch's default equals it, and is slower)",
        "host_cpu": host_cpu_model(),
        "host_logical_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"oracle.rife_oracle.ifnet47_forward (torch-CPU fp32 restatement, bit-exact vs the reference's IFNet('4.7') "
                  f"in the build container) on frames 0 and 1 of the GPU leg's own clip (identical tensors), 1 pair {H}x{W}, t = 0.5, run BEFORE the "
                  f"GPU leg; per thread count 1 warm-up + 3 timed forwards (median), (threads, median s/frame) tried: {tried}; best reported",
    }, refs


def main_single_process(args):
    """--gpus N without a torch.distributed launcher: ONE process drives N devices (multidev.py) — one host thread and stream
    per device, weights packed on device 0 and broadcast as one flat buffer over RCCL (ncclCommInitAll clique), every device
    runs its own B-pair stream (weak scaling) and the new frames are all-gathered in place by grouped per-root broadcasts on
    per-device communication streams, overlapped with the next step.  Same timed region as the multi-process form: all devices
    synchronised + a barrier on both sides, so the elapsed time is the slowest device's."""
    import threading

    import __graft_entry__ as ge

    ge.build()
    ge.load_package()
    from cfi_amd import _lib, multidev, synth

    N, B, H, W, K, Wm = args.gpus, args.batch, args.height, args.width, args.steps, args.warmup
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    if torch.cuda.device_count() < N:
        raise SystemExit(f"--gpus {N}: only {torch.cuda.device_count()} device(s) visible")
    devices = list(range(N))
    sd = synth.rife47_synth_state_dict(1234)
    torch.cuda.set_device(0)
    reserve = 0
    if N > 1 and not args.no_gather and args.reserve_cus > 0 and _lib.load().vfi_comm_all_gather_mode() == 1:
        # (RCCL is bound at the first communicator: the cap must be in the environment before it; see main() for the reasoning)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(args.reserve_cus))
        try:
            reserve = max(0, int(os.environ["NCCL_MAX_NCHANNELS"]))
        except ValueError:
            reserve = args.reserve_cus
        _lib.check(_lib.load().vfi_set_reserved_cus(reserve), "vfi_set_reserved_cus")
    group = multidev.RifeDeviceGroup(sd, "4.7", devices)
    comm = group.comm if group.comm is not None else multidev.Comm(devices)      # N = 1: the degenerate clique, same calls
    gather = not args.no_gather
    n_clip = clip_frames(B)
    raw, bufs = [None] * N, [None] * N
    for r in devices:
        torch.cuda.set_device(r)
        group.engines[r].configure(H, W, B, B + 1, 1.0)
        g = torch.Generator(device="cpu").manual_seed(r)
        raw[r] = torch.rand((n_clip, H, W, 3), generator=g, dtype=torch.float32).to(f"cuda:{r}")
        bufs[r] = [torch.empty((N * B, H, W, 3), dtype=torch.float32, device=f"cuda:{r}") for _ in range(2)]
    torch.cuda.set_device(0)
    slot0, slot1, ts = list(range(B)), list(range(1, B + 1)), [0.5] * B
    barrier = threading.Barrier(N)
    ev_done = [None] * N
    ev_gath = [[None, None] for _ in range(N)]
    per = B * H * W * 3
    times = {}
    errors = []

    def step(r, i, main):
        k = i & 1
        eng = group.engines[r]
        if ev_gath[r][k] is not None:
            main.wait_event(ev_gath[r][k])           # this buffer's previous all-gather has finished
        base = clip_base(B, k)
        eng.load_frames(list(range(B + 1)), [raw[r][base + j] for j in range(B + 1)])
        eng.interpolate(slot0, slot1, ts, bufs[r][k][r * B:(r + 1) * B])
        if gather:
            e = torch.cuda.Event()
            e.record(main)
            ev_done[r] = e
            barrier.wait()                           # every device has enqueued step i
            if r == 0:                               # one thread issues the collective for all devices
                for d in devices:
                    with torch.cuda.device(d):
                        comm.streams[d].wait_event(ev_done[d])
                comm.all_gather_v([bufs[d][k].data_ptr() for d in devices], [per] * N)
                for d in devices:
                    with torch.cuda.device(d):
                        g_ = torch.cuda.Event()
                        g_.record(comm.streams[d])
                        ev_gath[d][k] = g_
            barrier.wait()                           # the collective of step i is enqueued before anyone reuses its events

    def worker(r):
        try:
            torch.cuda.set_device(r)
            main = torch.cuda.current_stream()
            for i in range(Wm):
                step(r, i, main)
            torch.cuda.synchronize(r)
            barrier.wait()
            if r == 0:
                times["t0"] = time.perf_counter()
            for i in range(K):
                step(r, i, main)
            torch.cuda.synchronize(r)                # compute and communication streams of this device
            barrier.wait()
            if r == 0:
                times["t1"] = time.perf_counter()
        except BaseException as e:  # noqa: BLE001
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in devices[1:]]
    for t in threads:
        t.start()
    worker(0)
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    elapsed = times["t1"] - times["t0"]

    # roofline leg: the dominant kernel's launch duration by HIP events on device 0's stream (the library's tracer is
    # process-wide, so this pass runs device 0 alone: K steps, no collective)
    torch.cuda.set_device(0)
    lib = _lib.load()
    lib.vfi_trace_reset()
    lib.vfi_trace_enable(1)
    eng0 = group.engines[0]
    t0 = time.perf_counter()
    for i in range(K):
        eng0.load_frames(list(range(B + 1)), [raw[0][clip_base(B, i & 1) + j] for j in range(B + 1)])
        eng0.interpolate(slot0, slot1, ts, bufs[0][i & 1][:B])
    torch.cuda.synchronize(0)
    traced = time.perf_counter() - t0
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    lib.vfi_trace_reset()
    mode = "RCCL grouped per-root broadcasts" if lib.vfi_comm_all_gather_mode() == 1 else "direct peer copies, one per ordered device pair"
    res = result_line(args, N, elapsed, traced, rep, eng0, f"all_gather_v ({mode}; in place, comm streams), overlapped" if gather else "none")
    res["config"]["launch"] = f"one process, {N} device thread(s) (multidev.py; weights broadcast over RCCL)"
    res["config"]["reserved_cus"] = reserve
    _lib.load().vfi_set_reserved_cus(0)
    if not args.no_strong:
        try:
            res["strong_4k_x4"] = strong_4k_x4(args, torch.device("cuda", 0), N, 0, "nccl", group=group)
        except Exception as e:
            res["strong_4k_x4"] = {"error": f"{type(e).__name__}: {e}"}
    group.close()
    if N == 1 and group.comm is None:
        comm.close()
    print(json.dumps(res), flush=True)


def result_line(args, world, elapsed, traced, rep, eng, collective, clock=None):
    """The JSON line's common part (metric, roofline of the dominant kernel, per-kernel table).  ``clock``: {"timed": ..., "traced": ...,
    "sysfs": ...} from the clock probe (None where it was not run)."""
    B, H, W, K, Wm = args.batch, args.height, args.width, args.steps, args.warmup
    conv_flop, _ = eng.work_per_task()
    hp, wp = -(-H // 64) * 64, -(-W // 64) * 64
    dom = "resconv_c64"
    calls, ms = rep.get(dom, (0, 0.0))
    flop_per_launch = 2.0 * B * (hp // 4) * (wp // 4) * 64 * 64 * 9
    avg_ms = ms / calls if calls else float("nan")
    achieved = flop_per_launch / (avg_ms * 1e-3) / 1e12 if calls else float("nan")
    from cfi_amd import _lib

    # the product library has no algorithm switch (test taps are compiled out: include/vfi_hip_test.h): Winograd wherever the rule allows
    wino = not (_lib.is_test_build() and _lib.load().vfi_test_conv_algo(-1) == 1)
    exec_div = 2.25 if wino else 1.0
    # HBM traffic of the dominant kernel comes from PMC counters, which need their own rocprofv3 passes (--pmc cannot share a run
    # with this timing): the figure is the one tools/profile_round.sh measured for THIS kernel at the recorded batch, scaled to the
    # batch of this run, and `traffic_source` says so.  None when no measurement of the kernel in use is on file.
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = dom + ("_winograd" if wino else "")
            if tj.get(key) is not None:   # measured per launch at the batch recorded in the file; linear in the batch
                traffic = int(tj[key] * B / float(tj.get("_detail", {}).get("batch", B)))
                traffic_src = f"profiles/roofline_traffic.json ({tj.get('_detail', {}).get('source', 'rocprofv3 --pmc passes')}), not measured in this run"
        except Exception:
            traffic = None
    # HBM-class kernels of the RIFE step (algorithmic bytes per task; `full` = padded pixels): the flow F is
    # 16 B per full-resolution pixel (read + written), a block output T 2 planes x 16 B per block-resolution pixel, a warp touches
    # both frame packs (2 frames x (image + feature plane) x 16 B) at the pixels the next block keeps (all of them at scales 1 and 2,
    # the centre 2x2 of every 4x4 cell at scale 4), the next block's input X is 24 channels x 4 B per pixel of its resolution
    full = float(hp * wp)
    hbm_bytes = {
        "stage_trans4": ("block 0->1 transition (no previous flow): T(1/8) in, warps on 4/16 of the pixels, F out, X(1/4) out",
                         (32 / 64 + 64 * 4 / 16 + 16 + 96 / 16) * full),
        "stage_trans2": ("block 1->2 transition: T(1/4) + F in, warps, F out, X(1/2) out", (32 / 16 + 16 + 64 + 16 + 96 / 4) * full),
        "trans1_conv0a": ("block 2->3 transition fused into block 3's conv0.0: T(1/2) + F in, warps, F out, A0 (32 ch at 1/2) out; X never stored",
                          (32 / 4 + 16 + 64 + 16 + 128 / 4) * full),
        "final_blend": ("last warp x2 + sigmoid blend + crop + clamp: T + F + image planes in, RGB frame out", (32 + 16 + 32) * full + 12.0 * H * W),
    }
    # the frame pack (clamp / pad + encode.0 + encode.1 in one kernel): per FRAME, RGB fp32 in, the two planar4 planes out; B + 1 frames per step
    enc_frame_bytes = 12.0 * H * W + 32.0 * full
    roofline_hbm = []
    hbm_traffic, hbm_traffic_src = {}, None      # PMC bytes per launch of these kernels from the committed passes (they cannot share a run with this timing)
    try:
        hk = json.load(open(tpath))["hbm_kernels"]
        hbm_traffic = {k: v * B / float(hk["batch"]) * (hp * wp) / float(hk["padded_pixels"]) for k, v in hk["bytes_per_launch"].items()}
        hbm_traffic_src = "profiles/roofline_traffic.json (" + hk["source"] + "), not measured in this run"
    except Exception:  # noqa: BLE001
        pass

    def with_traffic(e, name):
        if name in hbm_traffic:
            e["traffic"] = int(hbm_traffic[name])
            e["traffic_over_algorithmic"] = round(e["traffic"] / e["algorithmic_bytes_per_launch"], 3)
            e["traffic_source"] = hbm_traffic_src
        return e

    for name, (what, per_task) in hbm_bytes.items():
        if name in rep and rep[name][0]:
            c, m = rep[name]
            roofline_hbm.append(with_traffic(hbm_entry(f"{name}: {what}; {B} tasks per launch", per_task * B, m / c, c), name))
    if "encode_batch" in rep and rep["encode_batch"][0]:
        c, m = rep["encode_batch"]
        e = hbm_entry(f"encode_batch: frame pack (clamp / pad + encode.0 Conv 3->16 s2 + encode.1 Deconv 16->4) of {B + 1} frames in one persistent "
                      f"launch: RGB fp32 in, planar4 image + feature planes out", enc_frame_bytes * (B + 1), m / c, c)
        e["valu_floor_ms"] = round((B + 1) * (hp * wp / 1024.0) * 256 * (2 * 27 * 16 + 1024) / VALU_LANE_OPS_PER_S * 1e3, 4)      # per 32x32 tile: 2 rounds x 432 + 1024 FMA per thread
        if "encode_batch" in hbm_traffic:      # (recorded for B + 1 = 33 frames)
            hbm_traffic["encode_batch"] = hbm_traffic["encode_batch"] * (B + 1) / float(B) * 32.0 / 33.0
        roofline_hbm.insert(0, with_traffic(e, "encode_batch"))
    elif "encode_fused" in rep and rep["encode_fused"][0]:
        c, m = rep["encode_fused"]
        roofline_hbm.insert(0, hbm_entry("encode_fused: frame pack of ONE frame per launch", enc_frame_bytes, m / c, c))
    total_ms = sum(v[1] for v in rep.values())
    kernels = {k: {"calls": v[0], "ms": round(v[1], 3), "share": round(v[1] / total_ms, 4)} for k, v in rep.items()}
    # the clock the chip sustained inside the dominant kernel: timed region (what `value` ran at) and traced pass (what avg_launch_ms ran at)
    clock_obj, frac_at_clock, mhz_traced = None, None, None
    if clock and clock.get("traced"):
        ct, cm = clock["traced"], clock.get("timed")
        mhz_traced = ct["shader_mhz"]
        frac_at_clock = round(achieved / exec_div / (PEAK_FP32_MFMA_TFLOPS * mhz_traced / 2400.0), 4) if calls else None
        clock_obj = {
            "shader_mhz": (cm or ct)["shader_mhz"],
            "shader_mhz_min_max": [(cm or ct)["min"], (cm or ct)["max"]],
            "region": "timed region" if cm else "traced pass",
            "traced_pass_shader_mhz": ct["shader_mhz"],
            "all_winograd_launches_mhz": (cm or ct)["all_winograd_launches_mhz"],
            "kernel": dom,
            "launches": (cm or ct)["launches"],
            "source": "s_memtime (shader cycles) / s_memrealtime (100 MHz) deltas of workgroup 0 of every launch of the dominant kernel (vfi_clock_probe); "
                      "peak clock 2400 MHz (MI355X_MICROARCH.md)",
            # s_memrealtime ticks per microsecond of HIP-event time of the same launches: 100 if the constant-rate counter runs at 100 MHz
            # (workgroup 0 lives a little shorter than the launch, so slightly below)
            "realtime_ticks_per_event_us": round(ct["avg_ticks_per_launch"] / (avg_ms * 1e3), 3) if calls and avg_ms else None,
            "sysfs": clock.get("sysfs"),
        }
    return {
        # BASELINE.json's metric.  ONE meaning for `value` at every N (the bench contract's): the WHOLE-JOB aggregate over n_gpus — the
        # driver divides by N itself; the per-GPU rate BASELINE.json's metric name speaks of is `per_gpu_frames_per_s` (== value at N = 1)
        "metric": "interpolated frames/sec/GPU @1080p RIFE4.7 2x" + ("" if world == 1 else f" (value = aggregate over {world} GPUs; per GPU: per_gpu_frames_per_s)"),
        "value": round(world * B * K / elapsed, 3),
        "unit": "frames/s" if world == 1 else f"frames/s, whole job = sum over {world} GPUs (weak scaling: {B} pairs/step/GPU)",
        "per_gpu_frames_per_s": round(B * K / elapsed, 3),
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": round(elapsed / K * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"RIFE 4.7 2x, {H}x{W} synthetic frame-pair stream, {B} pairs/step/GPU resident in HBM "
                        f"(BASELINE.json configs[1]; SURVEY 8d config 2 clip: {clip_frames(B)} frames torch.manual_seed(0) torch.rand); "
                        f"seeded random-init weights",
            "pairs_per_step_per_gpu": B,
            "per_gpu_frames_per_s": round(B * K / elapsed, 3),
            "new_frame_collective": collective,
            "target_frames_per_s_per_gpu": 30,
        },
        "roofline": {
            "kernel": ("conv_wino_kernel<8> (Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32)" if wino else "conv_mfma2_kernel<s1,3x3> (direct implicit GEMM)")
                      + " as resconv_c64 (block3 ResConv 64->64 @%dx%d, batch %d)" % (hp // 4, wp // 4, B),
            "bound": "mfma",
            # hardware utilisation: MFMA FLOP the kernel ISSUES per launch / launch duration / peak  (<= 1)
            "achieved": round(achieved / exec_div, 3),
            "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / exec_div / PEAK_FP32_MFMA_TFLOPS, 4),
            # the same launches against the peak AT THE CLOCK THEY RAN AT (64 FLOP/clk/SIMD x 1024 SIMDs x measured MHz): what the kernel
            # makes of the cycles it is given; `frac` additionally carries the box's clock / power state
            "frac_at_clock": frac_at_clock,
            "clock_mhz": mhz_traced,
            "flop_per_launch": flop_per_launch / exec_div,
            "flop_definition": ("executed: the Winograd F(2x2,3x3) form issues 16 MFMA multiplications per 2x2 output tile and channel pair instead "
                                "of 36 = direct / 2.25, on whole 16x8-pixel regions") if wino else "direct form: every algorithmic FLOP is an MFMA FLOP",
            "traffic": traffic,
            "traffic_source": traffic_src,
            "launches": calls,
            "avg_launch_ms": round(avg_ms, 4),
            "algorithmic_equiv": {
                "note": "the same launch priced in direct-form FLOP (SURVEY 8d: 2 * pixels * Cin * Cout * 9 — what the layer is worth to any "
                        "implementation); > peak because Winograd skips 5/9 of the multiplications, NOT a roofline fraction",
                "flop_per_launch": flop_per_launch,
                "achieved": round(achieved, 3),
                "x_peak": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
            },
            "traced_ms_per_step": round(traced / K * 1e3, 3),
        },
        "clock": clock_obj,
        "roofline_hbm": roofline_hbm,
        "conv_tflops_whole_net": round(conv_flop * B * K / elapsed / 1e12, 3),
        "kernels": kernels,
    }


def extras_watchdog(res, rank, deadline_s):
    """Timer for the legs AFTER the timed region of a multi-process run: when it fires, rank 0 prints the headline line it
    already holds (plus a note saying which legs are missing) and every rank leaves with os._exit — the only exit that works
    while the main thread sits inside a collective."""
    import threading

    def bail():
        if rank == 0 and res is not None:
            res["incomplete"] = True
            res.setdefault("notes", []).append(f"legs after the timed region did not finish within {deadline_s:.0f} s; line printed by the watchdog")
            print(json.dumps(res), flush=True)
        # the headline line above is valid and complete (`"incomplete": true` says that later legs are missing) — unless its parity gate failed
        os._exit(1 if res is not None and not res.get("parity", {}).get("ok", True) else 0)

    t = threading.Timer(deadline_s + (0.0 if rank == 0 else 5.0), bail)   # rank 0 first, so its line is out before peers drop
    t.daemon = True
    t.start()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="frame pairs per step per GPU (1..32)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle's timing (one forward still runs for the parity gate)")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity gate (and with it every oracle forward): profiling passes only")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive node leg")
    ap.add_argument("--e2e-long-frames", type=int, default=129, help="frames of the long-clip e2e leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the FILM / M2M device-resident numbers")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather of new frames")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling 4K x4 leg (BASELINE configs[3])")
    ap.add_argument("--peer-copy-leg", action="store_true",
                    help="N>1: also run the extra weak-scaling leg that exchanges frames by IPC-mapped peer copies instead of a collective (opt-in: "
                         "it has only ever run with its ranks on one GPU)")
    ap.add_argument("--reserve-cus", type=int, default=16,
                    help="N>1 over RCCL: compute units the persistent kernels leave to the overlapped all-gather's kernel (also caps "
                         "RCCL's channels to the same number unless NCCL_MAX_NCHANNELS is set); 0 = none")
    ap.add_argument("--extras-deadline", type=float, default=420.0,
                    help="multi-process runs: seconds the legs after the timed region may take before the headline line is printed without them")
    ap.add_argument("--strong-height", type=int, default=2160)
    ap.add_argument("--strong-width", type=int, default=3840)
    ap.add_argument("--strong-frames", type=int, default=17)
    ap.add_argument("--strong-reps", type=int, default=3)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) | gloo (plumbing test on one GPU)")
    ap.add_argument("--device-threads", action="store_true",
                    help="one process driving N devices with host threads (multidev.py) instead of one process per GPU: opt-in for any N")
    ap.add_argument("--dry-run-ranks", type=int, default=0,
                    help="plumbing check on ONE GPU: run the full N-rank control flow (weight broadcast, reserve trials, gather, strong leg, watchdog) "
                         "as N gloo ranks that all use device 0; the numbers in the line mean nothing")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.device_threads:
        return main_single_process(args)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.dry_run_ranks > 1):
        # One process per GPU over RCCL is THE multi-GPU path (the driver launches it that way itself); asked for N > 1 without a launcher,
        # become that launch.  --dry-run-ranks N: the same N ranks, gloo, all on device 0.
        import socket

        n = args.dry_run_ranks if args.dry_run_ranks > 1 else args.gpus
        argv, skip = [], 0
        for a in sys.argv[1:]:
            if skip:
                skip -= 1
            elif a in ("--gpus", "--dry-run-ranks", "--backend"):
                skip = 1
            elif not a.startswith(("--gpus=", "--dry-run-ranks=", "--backend=")):
                argv.append(a)
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__), "--gpus", str(n), "--backend", "gloo" if args.dry_run_ranks > 1 else args.backend] + argv
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the hot path has no CPU fallback)"
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)

    import torch.distributed as dist

    # The new-frame all-gather overlaps the next step's kernels.  RCCL's kernel stays resident for the whole collective and takes whole
    # compute units from the library's one-workgroup-per-CU kernels (+24 % per step with a 16-workgroup stand-in on one GPU,
    # profiles/r04_reserved_cus.txt); so RCCL is held to `reserve` channels (one workgroup each) and the library leaves as many
    # units free (vfi_set_reserved_cus below): +8.5 % instead.  A caller's own NCCL_MAX_NCHANNELS wins and sets the reserve.
    reserve = 0
    if world > 1 and not args.no_gather and args.reserve_cus > 0:
        reserve = args.reserve_cus
        if args.backend == "nccl":
            os.environ.setdefault("NCCL_MAX_NCHANNELS", str(args.reserve_cus))
            try:
                reserve = max(0, int(os.environ["NCCL_MAX_NCHANNELS"]))
            except ValueError:
                reserve = args.reserve_cus
        # (gloo plumbing runs take the same code path: the trials below are then between equals)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world,
                                device_id=dev if args.backend == "nccl" else None)
    if rank == 0:
        ge.build()  # no-op when the in-tree libvfi_hip.so is current
    if world > 1:
        dist.barrier()
    ge.load_package()
    from cfi_amd import _lib, synth
    from cfi_amd.dist import broadcast_state_dict
    from cfi_amd.rife import RifeEngine
    from cfi_amd.rife_spec import rife47_shapes

    B, H, W, K, Wm = args.batch, args.height, args.width, args.steps, args.warmup
    sd = synth.rife47_synth_state_dict(1234) if rank == 0 or world == 1 else None
    if world > 1:  # weights live on rank 0 and are broadcast once over RCCL (21.3 MB)
        sd = broadcast_state_dict(sd, rife47_shapes(), dev if args.backend == "nccl" else "cpu")
    eng = RifeEngine(sd, "4.7", device=dev)
    n_slots = B + 1
    eng.configure(H, W, B, max(n_slots, 2), 1.0)

    # synthetic stream, resident in HBM before timing: SURVEY 8(d) config 2's clip — 33 frames at the default batch (32 pairs =
    # the whole clip per step; 2B+1 frames walked in two halves for B <= 16), torch.manual_seed(0) torch.rand i.i.d. U[0,1)
    # (seed + rank for N>1)
    n_clip = clip_frames(B)
    g = torch.Generator(device="cpu").manual_seed(rank)
    raw_host = torch.rand((n_clip, H, W, 3), generator=g, dtype=torch.float32)
    raw = raw_host.to(dev)
    # The oracle on the pair the parity check will look at — task 0 of the LAST timed step — run BEFORE the GPU leg on the identical
    # tensors; at N = 1 the same forwards are the cpu_baseline sample.
    par_base = clip_base(B, (K - 1) & 1)
    par_slots = sorted({0, B // 2, B - 1})      # first / middle / last task of the 32-task launch
    cpu_base, oracle_frames = None, None
    if rank == 0 and not args.no_parity:
        try:
            cpu_base, oracle_frames = cpu_baseline(sd, [(raw_host[par_base + j], raw_host[par_base + j + 1]) for j in par_slots],
                                                   timing=(world == 1 and not args.no_cpu_baseline))
        except Exception as e:  # noqa: BLE001  (the line then says so: parity.error)
            cpu_base, oracle_frames = {"error": f"{type(e).__name__}: {e}"}, None
    del raw_host
    outs = [torch.empty((B, H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty((world * B, H, W, 3), dtype=torch.float32, device=dev) for _ in range(2)] \
        if world > 1 and not args.no_gather and args.backend == "nccl" else None
    pending = [None, None]
    slot0 = list(range(B))
    slot1 = list(range(1, B + 1))
    ts = [0.5] * B

    def step(i, gather=True):
        k = i & 1
        if pending[k] is not None:
            pending[k].wait()  # the buffer's previous all-gather must be done before it is overwritten
            pending[k] = None
        # Every frame of the step is prepared + encoded inside the timed region: B + 1 frames for B pairs (in a real clip the
        # first one would be the previous step's last and already resident: one frame more work than the node does).
        base = clip_base(B, k)
        eng.load_frames(list(range(B + 1)), [raw[base + j] for j in range(B + 1)])      # ONE frame-pack launch (vfi_rife_load_frames)
        eng.interpolate(slot0, slot1, ts, outs[k])
        if world > 1 and not args.no_gather and gather:
            if gathered is not None:
                pending[k] = dist.all_gather_into_tensor(gathered[k], outs[k], async_op=True)
            else:  # gloo plumbing mode: stage through the host
                lst = [torch.empty((B, H, W, 3)) for _ in range(world)]
                dist.all_gather(lst, outs[k].cpu())

    def drain():
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize()

    def timed(nsteps, gather=True):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(i, gather)
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    if reserve:
        _lib.check(_lib.load().vfi_set_reserved_cus(reserve), "vfi_set_reserved_cus")
    for i in range(Wm):
        step(i)
    drain()
    reserve_trials = None
    if reserve:
        # RCCL's real kernel has never run beside this library (no multi-GPU box was available): how many units it takes, and for how
        # long, is an estimate.  So the reserve is CHOSEN here, untimed, from what this node actually does: a few steps each with the
        # planned reserve, none, and twice as many; every rank sees the same all-reduced times and takes the same decision.
        reserve_trials = {}
        for cand in (reserve, 0, 2 * reserve):
            _lib.check(_lib.load().vfi_set_reserved_cus(cand), "vfi_set_reserved_cus")
            timed(1)
            t = torch.tensor([timed(3)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            reserve_trials[cand] = float(t.item()) / 3
        reserve = min(reserve_trials, key=lambda c: (reserve_trials[c], c))
        _lib.check(_lib.load().vfi_set_reserved_cus(reserve), "vfi_set_reserved_cus")
    # the timed region, with the clock probe installed (workgroup 0 of every Winograd launch stamps two counters: no launch, no sync,
    # no extra kernel) and a side thread reading sclk / power from sysfs where the box has them
    clock = {}
    probe = None
    try:
        probe = ClockProbe(dev, K * 40 + 8)
    except Exception as e:  # noqa: BLE001
        clock["error"] = f"{type(e).__name__}: {e}"
    with SysfsSampler(local_rank % ndev) as sampler:
        elapsed = timed(K)
    clock["sysfs"] = sampler.summary()
    if probe is not None:
        clock["timed"] = clock_summary(probe.finish(), "resconv_c64")
    # tasks 0 / B/2 / B-1 of the LAST timed step's launch
    last_frames = [outs[(K - 1) & 1][j].cpu() for j in par_slots] if rank == 0 and oracle_frames is not None else None
    gather_cost = None
    if world > 1:
        ctl_dev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if not args.no_gather:
            # what the new-frame all-gather costs a step, two ways (both untimed extras, max over ranks): the same K steps WITHOUT the
            # collective (exposed cost = with - without: what overlap does not hide, including the compute units RCCL's kernel takes),
            # and the collective alone, back to back on an idle device (its own duration over xGMI)
            t = torch.tensor([timed(K, gather=False)], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            no_gather = float(t.item())
            alone = None
            if gathered is not None:
                dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(K):
                    dist.all_gather_into_tensor(gathered[i & 1], outs[i & 1])
                torch.cuda.synchronize()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ctl_dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                alone = float(t.item())
            gather_cost = {"exposed": round((elapsed - no_gather) / K * 1e3, 3), "standalone": round(alone / K * 1e3, 3) if alone is not None else None,
                           "ms_per_step_without_gather": round(no_gather / K * 1e3, 3), "bytes_per_rank_per_step": B * H * W * 3 * 4,
                           "note": "exposed = ms_per_step - the same K steps without the collective; standalone = K all_gather_into_tensor calls back to back on an idle device"}

    # ---- roofline leg: same K steps with per-kernel HIP events recorded on the launch stream (and the clock probe again: the
    # event durations and the clock then belong to the same launches)
    lib = _lib.load()
    try:
        probe = ClockProbe(dev, K * 40 + 8)
    except Exception:  # noqa: BLE001
        probe = None
    lib.vfi_trace_reset()
    lib.vfi_trace_enable(1)
    traced = timed(K)
    lib.vfi_trace_enable(0)
    rep = _lib.trace_report()
    lib.vfi_trace_reset()
    if probe is not None:
        clock["traced"] = clock_summary(probe.finish(), "resconv_c64")
    lib.vfi_set_reserved_cus(0)       # the later legs gather after their compute, not beside it

    res = None
    if rank == 0:
        res = result_line(args, world, elapsed, traced, rep, eng,
                          "none" if world == 1 or args.no_gather else
                          ("all_gather(RCCL), overlapped" if gathered is not None else "all_gather(gloo, host)"), clock=clock)
        # ---- in-run parity gate: the timed workload itself against the oracle (per-pixel fp32 |d| <= 1e-3, north_star)
        if args.no_parity:
            res["parity"] = {"skipped": "--no-parity"}
        elif oracle_frames is None:
            res["parity"] = {"ok": False, "error": (cpu_base or {}).get("error", "no oracle frame")}
        else:
            per = [parity_of(g_, w_) for g_, w_ in zip(last_frames, oracle_frames)]
            res["parity"] = {"max_abs": max(p["max_abs"] for p in per), "mean_abs": sum(p["mean_abs"] for p in per) / len(per),
                             "n_over_1e-3": sum(p["n_over_1e-3"] for p in per), "values": sum(p["values"] for p in per), "tol": 1e-3,
                             "ok": all(p["ok"] for p in per), "slots": par_slots, "per_slot_max_abs": [p["max_abs"] for p in per]}
            res["parity"]["what"] = (f"tasks {par_slots} (first / middle / last) of the last timed step's {B}-task launch (pairs = clip frames {par_base}+j, {par_base}+j+1, "
                                     f"t = 0.5) vs oracle.rife_oracle.ifnet47_forward on the same host tensors, all {H}x{W}x3 values of each")
        if cpu_base is not None and "error" not in cpu_base:
            res["cpu_baseline"] = cpu_base
        res["config"]["launch"] = "one process per GPU (torch.distributed)" if world > 1 else "one process, one GPU"
        # what the communicator itself reports (not the --gpus argument): ranks and backend of the default process group
        res["n_ranks_seen_by_rccl"] = dist.get_world_size() if world > 1 else 1
        res["collective_backend"] = (dist.get_backend() + (" (= RCCL on ROCm)" if dist.get_backend() == "nccl" else "")) if world > 1 else "none (N = 1)"
        res["all_gather_ms_per_step"] = gather_cost
        if world > 1:
            res["config"]["reserved_cus"] = reserve
            res["config"]["nccl_max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS")
            res["config"]["reserved_cus_note"] = ("reserved_cus was chosen in-run from the untimed trials below (the same step loop); RCCL's channel cap is bound when the "
                                                  "communicator is created and stays at the planned reserve whatever the trials choose")
            if reserve_trials is not None:
                res["config"]["reserved_cus_trials_ms_per_step"] = {str(c): round(v * 1e3, 3) for c, v in reserve_trials.items()}
    # The headline is measured; every later leg is extra.  A leg that RAISES is recorded as an error string; a leg that STALLS
    # (a collective whose peer died) cannot be recovered from inside the process, so a watchdog prints the headline line as it
    # stands and ends the rank instead of losing it to the launcher's timeout.
    parity_failed = False
    guard = extras_watchdog(res, rank, args.extras_deadline) if world > 1 else None
    strong = None
    if not args.no_strong:       # every rank takes part (strong scaling over the ranks)
        try:
            strong = strong_4k_x4(args, dev, world, rank, args.backend)
        except Exception as e:
            strong = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            res["strong_4k_x4"] = strong
    if world > 1 and not args.no_gather and args.peer_copy_leg:
        try:
            pc = peer_copy_gather_leg(eng, raw, B, H, W, max(2, min(K, 10)), dev, world, rank, args.backend)
        except Exception as e:
            pc = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            res["weak_peer_copy_gather"] = pc
    if world > 1 and not args.no_extras:
        eng.release() if hasattr(eng, "release") else None
        try:
            dist_extras = other_paths_dist(dev, H, W, world, rank, args.backend)
        except Exception as e:
            dist_extras = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            res["other_paths"] = dist_extras
    if guard is not None:
        guard.cancel()
    if rank == 0:
        if world == 1:
            eng.close()
            if not args.no_e2e:
                res["e2e"] = e2e_leg(sd, dev, H, W)
                try:      # a real (long) clip beside SURVEY's 33-frame one
                    res["e2e"]["long_clip"] = e2e_leg(sd, dev, H, W, n_frames=args.e2e_long_frames, reps=3, with_u8=False)
                except Exception as e:  # noqa: BLE001  (never lose the line to an extra leg)
                    res["e2e"]["long_clip"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_extras:
                res["other_paths"] = other_paths(dev, H, W, parity=not args.no_parity)
                try:
                    res["other_paths"].update(other_nodes(dev, H, W, parity=not args.no_parity))
                except Exception as e:  # noqa: BLE001  (never lose the line to an extra leg)
                    res["other_paths"]["other_nodes_error"] = f"{type(e).__name__}: {e}"
        print(json.dumps(res), flush=True)
        parity_failed = not res.get("parity", {}).get("ok", True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and parity_failed:
        raise SystemExit("bench.py: the timed workload does not match the oracle within 1e-3 (see `parity` in the line above)")


if __name__ == "__main__":
    main()
