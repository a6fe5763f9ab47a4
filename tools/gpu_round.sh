#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_rife.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/gpu_tests.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-400
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:(v['ms']/v['calls']) for k,v in d['kernels'].items() if k in ('stage_trans','prep_frame','encode_conv','encode_deconv','final_blend','stage_in0')})
PY
