#!/bin/bash
# round 3ab: one scalar instruction between the MFMAs of the Winograd K loop (default) vs none (128)
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for v in 0 128; do
echo "== wino_bench ABLATE=$v"; VFI_WINO_ABLATE=$v timeout 240 python tools/wino_bench.py rife "512->512" "64->64" 2>&1 | grep "rife\|film"
done
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py -m gpu -x -q 2>&1 | tail -3
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03ab_bench.json; python - <<PY
import json
d = json.loads(open('gpurun_out/r03ab_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items() if 'resconv' in k})
PY
} 2>&1 | tee gpurun_out/r03ab.log | tail -70
