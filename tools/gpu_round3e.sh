#!/bin/bash
# round 3: Winograd integrated everywhere — op tests, the three main model suites, benches of RIFE / FILM / M2M
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== op tests"; timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -3
echo "== wino_bench"; timeout 240 python tools/wino_bench.py "rife" 2>&1 | grep -v "Warning\|amdgpu.ids\|hipcc"
echo "== film + m2m + bocchi tests"; timeout 900 python -m pytest tests/test_gpu_film.py tests/test_gpu_m2m.py tests/test_gpu_bocchi.py -q -m gpu -x 2>&1 | tail -5
echo "== bench"; timeout 400 python bench.py --steps 5 --warmup 2 --no-e2e --no-cpu-baseline 2>&1 | grep -v "Warning\|amdgpu.ids" > gpurun_out/r03e_bench.json; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03e_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'])
print({k: round(v['ms'] / d['steps'], 3) for k, v in d['kernels'].items()})
print(d.get('other_paths'))
PY
echo "== film bench"; VFI_TRACE_SHAPES=1 timeout 200 python tools/film_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -60
echo "== m2m bench"; timeout 200 python tools/m2m_bench.py 2>&1 | grep -v "Warning\|amdgpu.ids" | head -40
} 2>&1 | tee gpurun_out/r03e.log | tail -150
