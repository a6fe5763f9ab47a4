"""CPU: the pure helpers behind bench.py's r5 objects (parity gate, clock summary, traffic scaling) and the goldens' host signature."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import golden_stats  # noqa: E402


def test_parity_of_is_the_per_pixel_gate():
    want = torch.rand(8, 9, 3)
    got = want.clone()
    got[1, 2, 0] += 9.0e-4
    p = bench.parity_of(got, want)
    assert p["ok"] and p["n_over_1e-3"] == 0 and abs(p["max_abs"] - 9.0e-4) < 1e-6 and p["values"] == 8 * 9 * 3
    got[3, 3, 1] += 1.1e-3
    p = bench.parity_of(got, want)
    assert not p["ok"] and p["n_over_1e-3"] == 1
    got[0, 0, 0] = float("nan")          # a NaN must fail the gate, not slip through a comparison
    assert not bench.parity_of(got, want)["ok"]


def test_clock_summary_from_probe_records():
    # (cycles, 100 MHz ticks) per launch: 2.3 GHz and 2.2 GHz launches of the dominant kernel, one of another kernel
    recs = {"resconv_c64": [(2_300_000, 100_000), (2_200_000, 100_000)], "lastconv_b3": [(4_000_000, 200_000)]}
    c = bench.clock_summary(recs, "resconv_c64")
    assert c["shader_mhz"] == 2250.0 and c["min"] == 2200.0 and c["max"] == 2300.0 and c["launches"] == 2
    assert abs(c["all_winograd_launches_mhz"] - (2300 + 2200 + 2000) / 3) < 0.1 and c["avg_ticks_per_launch"] == 100_000
    assert bench.clock_summary({"lastconv_b3": [(1, 1)]}, "resconv_c64") is None


def test_traffic_file_is_consistent_with_its_sources():
    j = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
    d = j["_detail"]
    assert j["resconv_c64_winograd"] == int(d["FETCH_SIZE_KB_avg"] * 1024 * 2 + d["WRITE_SIZE_KB_avg"] * 1024)      # x2: the gfx950 FETCH_SIZE unit
    assert 1.0 < j["resconv_c64_winograd"] / d["algorithmic_bytes"] < 1.3
    hk = j["hbm_kernels"]
    assert hk["batch"] == 32 and hk["padded_pixels"] == 1088 * 1920 and set(hk["bytes_per_launch"]) == {"trans1_conv0a", "stage_trans2", "stage_trans4", "final_blend", "encode_batch"}
    m = j["m2m_softsplat_sum"]
    assert m["algorithmic_bytes"] == 8 * 40 * 1088 * 1920 and 1.0 < m["bytes_per_launch"] / m["algorithmic_bytes"] < 2.5


def test_golden_tolerance_follows_the_host_signature(tmp_path):
    sig, details = golden_stats.host_signature()
    assert len(sig) == 16 and details["torch"] == torch.__version__
    same = tmp_path / "same.json"
    golden_stats.write_host_signature(str(same))
    assert golden_stats.golden_tol(str(same)) == 0.0                       # written here: a bit-exact pin
    other = tmp_path / "other.json"
    other.write_text(json.dumps({"signature": "0" * 16, "details": {}}))
    assert golden_stats.golden_tol(str(other)) == golden_stats.CROSS_HOST_TOL      # written elsewhere: the cross-host spread
    assert golden_stats.golden_tol(str(tmp_path / "missing.json")) == golden_stats.CROSS_HOST_TOL
    assert golden_stats.CROSS_HOST_TOL <= 2e-4 < 1e-3
