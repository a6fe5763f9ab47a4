#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --batch 8 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('value', d['value'], 'roofline', d['roofline']['achieved'])
"
done
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py tests/test_gpu_film.py tests/test_gpu_m2m.py -x -q -m gpu -k "not 4k and not config2 and not 1080p" 2>&1 | tail -2
