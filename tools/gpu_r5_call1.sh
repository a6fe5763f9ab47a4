#!/bin/bash
# round 5, GPU call 1: the self-verifying bench line, the new parity sizes, the Winograd cycle ledger, the M2M PMC passes
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
(time timeout 1200 python -m pytest tests/test_gpu_bench_line.py tests/test_gpu_m2m_ops.py tests/test_capi_symbols.py -x -q -m gpu) > gpurun_out/r05a_tests.log 2>&1
tail -6 gpurun_out/r05a_tests.log
timeout 400 python tools/wino_ledger.py > gpurun_out/r05_wino_cycle_ledger.txt 2>&1
tail -40 gpurun_out/r05_wino_cycle_ledger.txt
(time timeout 900 python bench.py) > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
tail -3 gpurun_out/r05a_bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05a_bench.json") if l.startswith("{")][0])
    print({k: d.get(k) for k in ("value", "ms_per_step", "parity", "clock")})
    print("roofline", {k: d["roofline"].get(k) for k in ("frac", "frac_at_clock", "clock_mhz", "avg_launch_ms")})
    print("cpu_baseline", d.get("cpu_baseline", {}).get("value"), "e2e", d.get("e2e", {}).get("value"))
except Exception as e:
    print("bench line unreadable:", e)
PY
ls /sys/class/drm/ 2>/dev/null | head; ls /sys/class/drm/card*/device/hwmon/*/ 2>/dev/null | head -30
bash tools/profile_m2m_r05.sh r05 2>&1 | tail -60
