#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_lds
timeout 240 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES -d gpurun_out/prof_lds -o lds -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8 > gpurun_out/prof_lds.log 2>&1
python tools/rocprof_summary.py pmc gpurun_out/prof_lds > gpurun_out/r01c_pmc_LDS_planar.txt 2>&1
head -9 gpurun_out/r01c_pmc_LDS_planar.txt
rm -rf gpurun_out/prof_lds
