#!/bin/bash
# rocprofv3 evidence for bench.py's configuration: kernel stats, then PMC passes (one counter group per pass; never
# combined with other trace domains).  Summaries are written under gpurun_out/prof_*/ and copied to profiles/ by hand.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
TAG=${1:-r01}
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-e2e --no-extras --no-strong"
run() {  # name, rocprof args...
  local name=$1; shift
  rm -rf gpurun_out/prof_$name
  timeout 240 rocprofv3 "$@" -d gpurun_out/prof_$name -o $name -- $CMD > gpurun_out/prof_$name.log 2>&1
  echo "$name rc=$?"
}
run stats --kernel-trace --stats
python tools/rocprof_summary.py stats "gpurun_out/prof_stats/*/*_results.db" > gpurun_out/${TAG}_kernel_stats.txt 2>&1 || python tools/rocprof_summary.py stats "gpurun_out/prof_stats/*_results.db" > gpurun_out/${TAG}_kernel_stats.txt 2>&1
# the Winograd ResConv launches of one step, split by trunk width (same kernel instantiation, dispatch order 8 x c192, c128, c96, c64)
PH="resconv_c192*8,resconv_c128*8,resconv_c96*8,resconv_c64*8"
python tools/rocprof_summary.py phases "gpurun_out/prof_stats" "conv_wino_kernel<8, 0, 0, 0>" "$PH" >> gpurun_out/${TAG}_kernel_stats.txt 2>&1
head -32 gpurun_out/${TAG}_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY"; do
  n=$(echo $c | tr ' ' '+' | cut -c1-40)
  run pmc_$n --kernel-trace --pmc $c
  python tools/rocprof_summary.py pmc "gpurun_out/prof_pmc_$n/*/*_results.db" > gpurun_out/${TAG}_pmc_$n.txt 2>&1 || python tools/rocprof_summary.py pmc "gpurun_out/prof_pmc_$n/*_results.db" > gpurun_out/${TAG}_pmc_$n.txt 2>&1
  for cn in $c; do python tools/rocprof_summary.py phases "gpurun_out/prof_pmc_$n" "conv_wino_kernel<8, 0, 0, 0>" "$PH" $cn >> gpurun_out/${TAG}_pmc_$n.txt 2>&1; done
  head -12 gpurun_out/${TAG}_pmc_$n.txt; tail -16 gpurun_out/${TAG}_pmc_$n.txt
done
rm -rf gpurun_out/prof_*/   # databases are large; the text summaries are what is kept
