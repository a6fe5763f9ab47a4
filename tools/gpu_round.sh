#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/film_bench.py 270 480 --check 2>&1 | grep -E "FILM|calls" | tee gpurun_out/film_bench_270.log
timeout 900 python tools/film_bench.py 1080 1920 --check 2>&1 | grep -E "FILM|calls" | tee gpurun_out/film_bench_1080.log
