#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1; tail -5 gpurun_out/r05_profile_round.log
(time timeout 900 python bench.py) > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r05_bench.json") if l.startswith("{")][0])
print({k: d.get(k) for k in ("value", "ms_per_step")}, d["parity"]["max_abs"], d["clock"]["shader_mhz"], d["roofline"]["frac"], d["roofline"]["frac_at_clock"], d["e2e"]["value"])
PY
ls gpurun_out/r05_*
