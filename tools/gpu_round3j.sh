#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== fuzz + rife batch + multidev + dist"; timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_multidev.py tests/test_gpu_dist_nodes.py "tests/test_gpu_rife.py::test_batch_invariance_and_determinism" -q -m gpu 2>&1 | tail -30
echo "== bench"; timeout 600 python bench.py --steps 5 --warmup 2 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03j_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03j_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print(json.dumps(d['roofline'], indent=0)[:1500])
for e in d['roofline_hbm'] + d['other_paths'].get('roofline_hbm', []): print(e['kernel'][:60], e['avg_launch_ms'], e['achieved'], e['frac'], e['frac_of_copy_rate'])
print(d['e2e']['value'], d['e2e']['seconds'], d['e2e']['uint8_clip']['value'])
print(d['other_paths']['film_2x'], d['other_paths']['m2m'])
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
} 2>&1 | tee gpurun_out/r03j.log | tail -70
