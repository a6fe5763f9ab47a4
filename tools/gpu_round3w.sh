#!/bin/bash
# round 3w: packed-fp32 output transform in the Winograd epilogue
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== wino_bench"; timeout 240 python tools/wino_bench.py rife "512->512" "64->64" m2m 2>&1 | grep "rife\|film\|m2m"
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py tests/test_gpu_film.py tests/test_gpu_m2m.py tests/test_gpu_ifrnet.py -m gpu -x -q 2>&1 | tail -3
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03w_bench.json; python - <<PY
import json
d = json.loads(open('gpurun_out/r03w_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items() if 'resconv' in k})
PY
} 2>&1 | tee gpurun_out/r03w.log | tail -70
