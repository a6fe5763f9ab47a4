#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/m2m_bench.py --check > gpurun_out/m2m_bench.log 2>&1
tail -60 gpurun_out/m2m_bench.log
timeout 300 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
tail -1 gpurun_out/bench.log | cut -c1-900
timeout 300 python -m pytest tests/test_gpu_m2m.py tests/test_gpu_rife.py -x -q -m gpu 2>&1 | tail -3
