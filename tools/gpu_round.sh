#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/r01d_pmc_waves.txt
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM"; do
rm -rf gpurun_out/prof_w
timeout 200 rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/prof_w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 8 > gpurun_out/prof_w.log 2>&1
python tools/rocprof_summary.py pmc gpurun_out/prof_w "conv_mfma2_kernel<1, 9, 2, 2" 2>&1 | tee -a gpurun_out/r01d_pmc_waves.txt | head -6
done
rm -rf gpurun_out/prof_w
