#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== op tests"; timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "winograd" 2>&1 | tail -3
echo "== wino_bench"; timeout 240 python tools/wino_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids\|hipcc"
echo "== bench"; timeout 400 python bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>&1 | grep -v "Warning\|amdgpu.ids" > gpurun_out/r03l_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03l_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['executed']['frac'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items()})
print(d.get('other_paths', {}).get('film_2x'), d.get('other_paths', {}).get('m2m'))
PY
} 2>&1 | tee gpurun_out/r03l.log | tail -40
