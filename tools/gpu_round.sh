#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py -x -q -m gpu -k "not 4k and not 40" 2>&1 | tail -2
for b in 16 8 1; do
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --batch $b 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch', d['config']['pairs_per_step_per_gpu'], 'value', d['value'], 'roofline', d['roofline']['achieved'])
"
done
