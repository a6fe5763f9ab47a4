#!/bin/bash
# round 3v: trans1_conv0a phase A in one pass (two pixels in flight on wave 0)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_rife.py -m gpu -x -q 2>&1 | tail -3
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03v_bench.json; python - <<PY
import json
d = json.loads(open('gpurun_out/r03v_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items()})
PY
} 2>&1 | tee gpurun_out/r03v.log | tail -70
