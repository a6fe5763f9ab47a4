"""-m gpu: bench.py's JSON line carries what the measurement contract asks for — a small configuration so
that the test takes seconds: `roofline.frac` is a hardware utilisation (<= 1) with the direct-form figure under
`algorithmic_equiv`, the frame pack is in `roofline_hbm`, the strong-scaling 4K x4 leg is present (here at a reduced size), and the
two-rank launch (gloo on the one GPU) shards that leg's task list unevenly."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--height", "272", "--width", "480", "--batch", "4", "--steps", "1", "--warmup", "1", "--no-e2e", "--no-cpu-baseline",
         "--strong-height", "136", "--strong-width", "240", "--strong-frames", "4", "--strong-reps", "1"]


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


def test_single_gpu_line(hip_lib):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = _line(r.stdout)
    rf = res["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] <= 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    eq = rf["algorithmic_equiv"]
    assert abs(eq["flop_per_launch"] / rf["flop_per_launch"] - 2.25) < 1e-9 and abs(eq["x_peak"] / rf["frac"] - 2.25) < 1e-2
    names = [e["kernel"].split(":")[0] for e in res["roofline_hbm"]]
    assert names[0] == "encode_batch" and {"stage_trans4", "stage_trans2", "trans1_conv0a", "final_blend"} <= set(names)
    for e in res["roofline_hbm"]:
        assert e["bound"] == "hbm" and 0 < e["frac"] <= 1.0
        assert 0.9 < e["traffic_over_algorithmic"] < 3.0, e          # r5: PMC bytes of the committed passes beside the algorithmic ones
    st = res["strong_4k_x4"]
    assert st["scaling"] == "strong" and st["n_gpus"] == 1 and st["tasks_per_rank"] == [9] and st["value"] > 0      # 3 pairs x 3 timesteps
    op = res["other_paths"]["roofline_hbm"]
    kinds = {e["bound"] for e in op if "bound" in e}
    assert kinds == {"hbm", "valu"}, op
    assert any("micro-benchmark" in e["kernel"] and "error" not in e for e in op)
    cv = [e for e in op if e.get("bound") == "valu"][0]
    assert 0 < cv["frac"] <= 1.0 and cv["valu_floor_ms"] > 0
    for k in ("gmfss_fortuna_union", "ifunet", "ifrnet_L"):       # r5: the SURVEY 8(f) nodes are in the driver-run line
        assert "error" not in res["other_paths"][k] and res["other_paths"][k]["conv_tflops_direct_form"] > 0, res["other_paths"][k]
    # r5: the line verifies itself — the timed workload against the oracle on identical tensors — and carries the shader clock
    par = res["parity"]
    assert par["ok"] and par["n_over_1e-3"] == 0 and 0 < par["max_abs"] <= 1e-3, par
    assert par["slots"] == [0, 2, 3] and par["values"] == 3 * 272 * 480 * 3 and len(par["per_slot_max_abs"]) == 3, par      # r6: first / middle / last task of the launch
    # r6: every leg that prints a number verifies it (FILM, M2M, the strong 4K x4 leg), and the multi-GPU keys exist at N = 1 too
    for leg in (res["other_paths"]["film_2x"], res["other_paths"]["m2m"], st):
        assert leg["parity"]["ok"] and leg["parity"]["values"] > 0 and "what" in leg["parity"], leg["parity"]
    # late r6: the SURVEY 8(f) legs verify their frames too (GMFSS only on a vector whose oracle conditioning is certified — the 1080p one of
    # the default run; at this test's size the object must exist and carry numbers), and every per-pair leg reports its pair lanes
    for k in ("ifunet", "ifrnet_L"):
        assert res["other_paths"][k]["parity"]["ok"] and res["other_paths"][k]["parity"]["n_over_1e-3"] == 0, res["other_paths"][k]["parity"]
    assert "max_abs" in res["other_paths"]["gmfss_fortuna_union"]["parity"], res["other_paths"]["gmfss_fortuna_union"]["parity"]
    for k in ("film_2x", "m2m", "gmfss_fortuna_union", "ifunet", "ifrnet_L"):
        pl = res["other_paths"][k]["pair_lanes"]
        assert "error" not in pl and pl["lanes"] >= 2 and pl["frames_per_s_2x"] > 0 and pl["vs_one_stream"] > 0.5, (k, pl)
    assert res["per_gpu_frames_per_s"] == res["value"] and res["n_ranks_seen_by_rccl"] == 1 and res["all_gather_ms_per_step"] is None
    assert "best of" in json.dumps(res.get("cpu_baseline", {"cores_policy": "best of"}))
    ck = res["clock"]
    assert ck and 500 < ck["shader_mhz"] <= 2500 and ck["launches"] >= 8 and ck["region"] == "timed region", ck
    assert 30 <= ck["realtime_ticks_per_event_us"] <= 101, ck          # s_memrealtime = 100 MHz (workgroup 0 lives a little shorter than the launch)
    assert rf["clock_mhz"] == ck["traced_pass_shader_mhz"] and 0 < rf["frac_at_clock"] <= 1.0
    assert abs(rf["frac_at_clock"] * rf["clock_mhz"] / 2400.0 - rf["frac"]) < 2e-3


def test_two_rank_line_shards_the_strong_leg(hip_lib):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-extras", "--peer-copy-leg"] + SMALL
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = _line(r.stdout)
    assert res["n_gpus"] == 2 and res["scaling"] == "weak"
    # r6: ONE meaning of `value` at N > 1 — the whole-job aggregate, said in `unit`; the per-GPU rate, the ranks the process group reports
    # and the all-gather's cost per step are top-level keys
    assert abs(res["value"] - 2 * res["per_gpu_frames_per_s"]) < 0.02 * res["value"] and "sum over 2 GPUs" in res["unit"], (res["value"], res["unit"])
    assert res["n_ranks_seen_by_rccl"] == 2 and "gloo" in res["collective_backend"]
    ag = res["all_gather_ms_per_step"]
    assert ag and ag["ms_per_step_without_gather"] > 0 and ag["bytes_per_rank_per_step"] == 4 * 272 * 480 * 3 * 4, ag
    assert res["strong_4k_x4"]["parity"]["ok"], res["strong_4k_x4"]["parity"]
    assert res["parity"]["ok"] and "cpu_baseline" not in res, res.get("parity")      # N > 1: the gate runs (one oracle forward on rank 0), the baseline does not
    st = res["strong_4k_x4"]
    assert st["tasks_per_rank"] == [5, 4] and st["n_gpus"] == 2 and st["value"] > 0, st
    # the reserve for an overlapped collective is chosen from untimed trials (planned, none, twice as many) and reported
    cfg = res["config"]
    assert set(cfg["reserved_cus_trials_ms_per_step"]) == {"16", "0", "32"} and cfg["reserved_cus"] in (0, 16, 32)
    assert all(v > 0 for v in cfg["reserved_cus_trials_ms_per_step"].values())
    # the extra weak-scaling leg that exchanges frames by IPC-mapped peer copies (no collective kernel): ran, validated what arrived
    pc = res["weak_peer_copy_gather"]
    assert "error" not in pc and pc["value"] > 0 and "fingerprints" in pc["validated"], pc


@pytest.mark.parametrize("n", [4, 8])
def test_dry_run_ranks(hip_lib, n):
    """the full N-rank control flow (weight broadcast, reserve trials, gather, strong leg with uneven blocks, the
    sharded FILM / M2M legs) for N = 4 and 8 as gloo ranks on the one GPU, launched by bench.py itself — so that the first run on an
    8-GPU node cannot die on plumbing.  The numbers mean nothing."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-ranks", str(n)] + SMALL, capture_output=True, text=True, timeout=1500, cwd=ROOT,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    res = _line(r.stdout)
    assert res["n_gpus"] == n and res["scaling"] == "weak" and res["value"] > 0 and not res.get("incomplete")
    assert res["parity"]["ok"]
    st = res["strong_4k_x4"]
    assert st["n_gpus"] == n and sum(st["tasks_per_rank"]) == 9 and len(st["tasks_per_rank"]) == n and st["value"] > 0, st
    assert set(res["config"]["reserved_cus_trials_ms_per_step"]) == {"16", "0", "32"}
    op = res["other_paths"]
    assert "error" not in op and op["film_2x"]["frames_per_s"] > 0 and op["m2m_2x"]["frames_per_s"] > 0, op
