#!/bin/bash
# Next-round starting point for the two-waves-per-SIMD Winograd kernel (docs/design/winograd.md section 6): SQ counter passes of
# conv_wino16_kernel next to conv_wino_kernel<8> on the block-3 RIFE layer (tools/wino_bench.py runs both), one counter group per
# pass, no tracing domains beside --kernel-trace.  ~25 s of GPU per pass.  Summaries land in gpurun_out/two_wave_pmc_*.txt.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
run() {  # tag, counters...
  local tag=$1; shift
  rm -rf gpurun_out/prof_tw_$tag
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" -d gpurun_out/prof_tw_$tag -o tw -- python tools/wino_bench.py "res_c64 x32" > gpurun_out/prof_tw_$tag.log 2>&1
  echo "$tag rc=$?"
  python tools/rocprof_summary.py pmc "gpurun_out/prof_tw_$tag/*/*_results.db" wino > gpurun_out/two_wave_pmc_$tag.txt 2>&1 || python tools/rocprof_summary.py pmc "gpurun_out/prof_tw_$tag/*_results.db" wino > gpurun_out/two_wave_pmc_$tag.txt 2>&1
  cat gpurun_out/two_wave_pmc_$tag.txt
  rm -rf gpurun_out/prof_tw_$tag/
}
SEL=${1:-all}
want() { [ "$SEL" = all ] || [ "$SEL" = "$1" ]; }
# what the waves wait for
want wait && run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS
# instruction mix actually issued (MFMA vs VALU vs LDS vs VMEM vs SALU)
want insts && run insts SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM
# LDS: conflicts, address stalls, data return
want lds && run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_LDS
# matrix pipe and issue
want mfma && run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
