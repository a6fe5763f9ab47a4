#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "=== pytest -m gpu"; timeout 900 python -m pytest tests -q -m gpu --no-header -rf 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
for b in 1 4 8 16; do timeout 300 python bench.py --steps 5 --warmup 2 --batch $b --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_b$b.json; python - <<PY
import json
d=json.load(open("gpurun_out/bench_b$b.json")); print("B=$b", d["value"], "fps", d["ms_per_step"], "ms/step", "resconv_c64", d["roofline"]["achieved"], "TF")
if $b in (1,8):
  for k,v in d["kernels"].items(): print("   %-16s %3d calls %8.3f ms/step %5.1f%%" % (k, v["calls"], v["ms"]/d["steps"], 100*v["share"]))
PY
done
