#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
(time timeout 2400 python -m pytest tests/test_gpu_bench_line.py -x -q -m gpu) > gpurun_out/r05b_tests.log 2>&1
tail -30 gpurun_out/r05b_tests.log
(time timeout 900 python bench.py) > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err
tail -3 gpurun_out/r05b_bench.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05b_bench.json") if l.startswith("{")][0])
    print({k: d.get(k) for k in ("value", "ms_per_step")}, d["parity"]["max_abs"], d["clock"]["shader_mhz"])
    op = d["other_paths"]
    print({k: op[k] for k in op if k != "roofline_hbm"})
    print("e2e", d["e2e"]["value"], d["e2e"]["seconds"], "strong", d["strong_4k_x4"]["value"], d["strong_4k_x4"]["seconds"])
except Exception as e:
    print("bench line unreadable:", e)
PY
