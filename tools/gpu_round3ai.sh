#!/bin/bash
# round 3ai: SQ counters of the one-wave and the two-wave Winograd kernels on the block-3 layer (one PMC pass, no tracing domains)
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/prof_w16
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/prof_w16 -o w16 -- python tools/wino_bench.py "res_c64 x32" > gpurun_out/prof_w16.log 2>&1
echo "rc=$?"
python tools/rocprof_summary.py pmc "gpurun_out/prof_w16/*/*_results.db" wino 2>&1 > gpurun_out/r03_pmc_two_wave_vs_one_wave.txt || python tools/rocprof_summary.py pmc "gpurun_out/prof_w16/*_results.db" wino > gpurun_out/r03_pmc_two_wave_vs_one_wave.txt 2>&1
cat gpurun_out/r03_pmc_two_wave_vs_one_wave.txt | head -30
rm -rf gpurun_out/prof_w16/
