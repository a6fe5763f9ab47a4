#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/gpu_tests.log
timeout 200 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('metric','value','n_gpus','steps','warmup','ms_per_step','dtype')}); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
