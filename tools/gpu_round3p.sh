#!/bin/bash
# round 3p: spread-DMA schedule of the Winograd kernel (VFI_WINO_ABLATE=1024) vs the boundary-issue schedule; HBM counter calibration
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== calibration plain"; timeout 60 tools/micro/hbm_known_traffic 1024
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/prof_cal_$c
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/prof_cal_$c -o cal -- tools/micro/hbm_known_traffic 1024 > gpurun_out/prof_cal_$c.log 2>&1
  echo "-- $c rc=$?"
  python tools/rocprof_summary.py pmc "gpurun_out/prof_cal_$c/*/*_results.db" 2>&1 || python tools/rocprof_summary.py pmc "gpurun_out/prof_cal_$c/*_results.db" 2>&1
done
rm -rf gpurun_out/prof_cal_*/
echo "== wino_bench boundary-issue"; timeout 240 python tools/wino_bench.py rife "512->512" "64->64" 2>&1 | grep "rife\|film"
echo "== wino_bench spread"; VFI_WINO_ABLATE=1024 timeout 240 python tools/wino_bench.py rife "512->512" "64->64" 2>&1 | grep "rife\|film"
echo "== parity spread"; VFI_WINO_ABLATE=1024 timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py -m gpu -x -q -k "wino or batch or golden or stage" 2>&1 | tail -4
echo "== parity default"; timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py -m gpu -x -q -k "wino or batch or golden or stage" 2>&1 | tail -4
for v in 0 1024; do
echo "== bench ABLATE=$v"; VFI_WINO_ABLATE=$v timeout 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03p_bench_$v.json; python - <<PY
import json
d = json.loads(open('gpurun_out/r03p_bench_$v.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items() if 'resconv' in k})
PY
done
} 2>&1 | tee gpurun_out/r03p.log | tail -70
