#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_rife.py tests/test_gpu_film.py -x -q -m gpu -k "not 4k and not config2" 2>&1 | tail -3
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --batch 8 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('value', d['value'], 'roofline', d['roofline']['achieved'], ' '.join(f\"{n}={v['ms']/v['calls']*1e3:.0f}us\" for n,v in k.items() if n.startswith(('resconv','conv0','lastconv'))))
"
