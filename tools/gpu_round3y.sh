#!/bin/bash
# round 3y: the N > 1 plumbing of bench.py on ONE GPU (two ranks, gloo) incl. the FILM / M2M distributed legs
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== bench 2 ranks gloo on one GPU"; HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --backend gloo --batch 8 2>/dev/null | grep '^{' > gpurun_out/r03y_bench_2ranks_gloo.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03y_bench_2ranks_gloo.json').read().strip().splitlines()[-1])
print('value', d['value'], 'n_gpus', d['n_gpus'], 'other_paths', json.dumps(d.get('other_paths')))
PY
} 2>&1 | tee gpurun_out/r03y.log | tail -30
