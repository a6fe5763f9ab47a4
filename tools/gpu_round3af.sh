#!/bin/bash
# round 3 final pass: full -m gpu suite (incl. the opt-in two-wave Winograd kernel's tests), bench.py default, RIFE with VFI_WINO_2WAVE=1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
echo "== bench default"; timeout 900 python bench.py 2>/dev/null | grep '^{' > gpurun_out/r03_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['executed']['frac'], d['roofline']['avg_launch_ms'], 'e2e', d['e2e']['value'], d['e2e']['uint8_clip']['value'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], 'film', d['other_paths']['film_2x']['ms_per_frame'], 'm2m', d['other_paths']['m2m']['prepare_ms_per_pair'])
PY
echo "== bench VFI_WINO_2WAVE=1"; VFI_WINO_2WAVE=1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | grep '^{' > gpurun_out/r03_bench_2wave.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03_bench_2wave.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items() if 'resconv' in k})
PY
} 2>&1 | tee gpurun_out/r03af.log | tail -30
