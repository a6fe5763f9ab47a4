#!/bin/bash
# bench.py per-kernel times under conv tile-variant overrides (VFI_VARIANT_OVERRIDE, by trace name)
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/variant_sweep.log
: > $OUT
run() {
  echo "=== override: $1" >> $OUT
  VFI_VARIANT_OVERRIDE="$1" timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernels']
        print('value', d['value'], ' '.join(f\"{n}={v['ms']/v['calls']*1e3:.0f}us\" for n,v in k.items() if n.startswith(('conv0','lastconv','resconv'))))
" >> $OUT 2>&1
}
run ""
for o in "$@"; do run "$o"; done
cat $OUT
