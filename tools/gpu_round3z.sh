#!/bin/bash
# round 3 evidence pass: full -m gpu suite, bench.py (default flags), rocprofv3 kernel stats + PMC passes
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
echo "== bench default"; timeout 900 python bench.py 2>/dev/null | grep '^{' > gpurun_out/r03_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['executed']['frac'], 'e2e', d['e2e']['value'], d['e2e']['uint8_clip']['value'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('other', json.dumps(d['other_paths'])[:400])
PY
echo "== profile"; bash tools/profile_round.sh r03
} 2>&1 | tee gpurun_out/r03z.log | tail -120
