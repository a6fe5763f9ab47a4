#!/bin/bash
# round 3x: where the node's end-to-end time goes now (device part 38 ms per 32 pairs) + FILM / M2M with the packed Winograd loop
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== e2e timeline"; VFI_HOST_TIMELINE=1 VFI_HOST_PROFILE=1 REPS=3 timeout 300 python tools/node_e2e.py 33 8 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy"
echo "== bench full"; timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "Warning\|amdgpu.ids\|Comfy" > gpurun_out/r03x_bench.json; python - <<PY
import json
d = json.loads(open('gpurun_out/r03x_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'])
print('e2e', json.dumps(d.get('e2e'))[:900])
print('other', json.dumps(d.get('other_paths'))[:1200])
PY
} 2>&1 | tee gpurun_out/r03x.log | tail -80
