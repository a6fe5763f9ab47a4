#!/bin/bash
# round 3r: fully-interior epilogue fast path; how much non-MFMA work one wave per SIMD hides under fp32 MFMAs
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== mfma_shadow"; timeout 120 tools/micro/mfma_shadow
echo "== wino_bench"; timeout 240 python tools/wino_bench.py rife "512->512" "64->64" 2>&1 | grep "rife\|film"
echo "== parity"; timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py -m gpu -x -q 2>&1 | tail -3
echo "== bench"; timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03r_bench.json; python - <<PY
import json
d = json.loads(open('gpurun_out/r03r_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items()})
PY
} 2>&1 | tee gpurun_out/r03r.log | tail -70
