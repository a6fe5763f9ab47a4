#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== rife tests"; timeout 900 python -m pytest tests/test_gpu_rife.py -q -m gpu -x 2>&1 | tail -4
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03n_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03n_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['executed']['frac'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items()})
print(d['e2e']['value'], d['e2e']['seconds'], d['e2e']['uint8_clip']['value'])
print(d.get('other_paths', {}).get('film_2x'), d.get('other_paths', {}).get('m2m'))
PY
} 2>&1 | tee gpurun_out/r03n.log | tail -40
