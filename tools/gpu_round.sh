#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for b in 2 4; do
timeout 200 python bench.py --height 2160 --width 3840 --batch $b --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('4K batch', d['config']['pairs_per_step_per_gpu'], 'value', d['value'], 'fps; ms/step', d['ms_per_step'], 'roofline', d['roofline']['achieved'], 'TF/s', 'whole-net', d['conv_tflops_whole_net'])
" | tee -a gpurun_out/bench_4k.log
done
