#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/prof_pmc_$c
  timeout 100 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/prof_pmc_$c -o pmc_$c -- $CMD > gpurun_out/prof_pmc_$c.log 2>&1
  echo "$c rc=$?"
  python tools/rocprof_summary.py pmc gpurun_out/prof_pmc_$c > gpurun_out/r01e_pmc_$c.txt 2>&1
  head -14 gpurun_out/r01e_pmc_$c.txt | cut -c1-150
  rm -rf gpurun_out/prof_pmc_$c
done
