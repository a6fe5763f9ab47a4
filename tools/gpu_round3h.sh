#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== bench"; timeout 400 python bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>&1 | grep -v "Warning\|amdgpu.ids" > gpurun_out/r03h_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03h_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items()})
print(d.get('other_paths'))
PY
echo "== rife tests"; timeout 600 python -m pytest tests/test_gpu_rife.py -q -m gpu -x 2>&1 | tail -3
} 2>&1 | tee gpurun_out/r03h.log | tail -30
