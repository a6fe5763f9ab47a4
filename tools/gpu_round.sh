#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 330 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 120 python bench.py 2>/dev/null | tail -1 | tee gpurun_out/final_bench.json
