#!/bin/bash
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python tools/film_algo_ab.py > gpurun_out/r05_film_algo_ab.txt 2>&1; grep -v amdgpu.ids gpurun_out/r05_film_algo_ab.txt | head -60
