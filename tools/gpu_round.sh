#!/bin/bash
export TMPDIR=/tmp
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']; print('value', d['value'], 'stage_trans', k['stage_trans']['ms']/k['stage_trans']['calls'], 'conv0a_b3', k['conv0a_b3']['ms']/k['conv0a_b3']['calls'])
"
