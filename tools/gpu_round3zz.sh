#!/bin/bash
# end-of-round evidence pass 2 (after the FILM window change): smoke, full -m gpu suite, bench.py default, the other nodes' benches
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -2
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4
echo "== bench default"; timeout 900 python bench.py 2>/dev/null | grep '^{' > gpurun_out/r03_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['executed']['frac'], d['roofline']['avg_launch_ms'], 'traffic', d['roofline']['traffic'], 'e2e', d['e2e']['value'], d['e2e']['uint8_clip']['value'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('other', json.dumps(d['other_paths'])[:330])
PY
echo "== ifrnet"; timeout 200 python tools/ifrnet_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -6
echo "== ifunet"; timeout 200 python tools/ifunet_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -6
echo "== gmfss"; timeout 300 python tools/gmfss_bench.py --coherent 2>&1 | grep -v "Warning\|amdgpu.ids" | head -8
echo "== m2m"; timeout 200 python tools/m2m_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -4
} 2>&1 | tee gpurun_out/r03zz.log | tail -60
