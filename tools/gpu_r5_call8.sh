#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
VFI_TRACE_SHAPES=1 timeout 600 python tools/film_bench.py > gpurun_out/r05_film_bench_shapes.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_film_bench_shapes.txt | cut -c1-200 | head -70
