#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/profile_round.sh r01d > gpurun_out/profile_round.log 2>&1
grep -A3 "^kernel" gpurun_out/r01d_kernel_stats.txt | head -5
head -3 gpurun_out/r01d_pmc_FETCH_SIZE.txt; head -3 gpurun_out/r01d_pmc_WRITE_SIZE.txt; head -4 "gpurun_out/r01d_pmc_SQ_VALU_MFMA_BUSY_CYCLES+SQ_BUSY_CYCLES+.txt"
timeout 200 python tools/m2m_bench.py 2>&1 | grep -E "prepare" | tail -1
timeout 200 python tools/film_bench.py 2>&1 | grep -E "ms per interpolated" | tail -1
timeout 200 python tools/rife_arch_bench.py 2>&1 | grep "^RIFE"
