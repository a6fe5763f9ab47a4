#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== wino_bench xcd on"; timeout 240 python tools/wino_bench.py rife "512->512" 2>&1 | grep "rife\|film"
echo "== wino_bench xcd off"; VFI_WINO_XCD=0 timeout 240 python tools/wino_bench.py rife "512->512" 2>&1 | grep "rife\|film"
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03o_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03o_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items() if 'resconv' in k})
PY
echo "== bench xcd off"; VFI_WINO_XCD=0 timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03o_bench2.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03o_bench2.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items() if 'resconv' in k})
PY
} 2>&1 | tee gpurun_out/r03o.log | tail -40
