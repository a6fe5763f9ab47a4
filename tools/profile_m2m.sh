#!/bin/bash
# rocprofv3 evidence for the CURRENT M2M kernels at 1080p (none existed since round 1): kernel stats, then FETCH_SIZE / WRITE_SIZE / wait
# counters in separate passes (--pmc never shares a run with a trace domain other than --kernel-trace), for tools/m2m_bench.py (the network:
# prepare + render, coherent flows) and for tools/splat_bench.py (SURVEY 8d config 5's incoherent splat field).  Text summaries only.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
TAG=${1:?usage: profile_m2m.sh <tag, e.g. r06>}
prof() {  # name, command, rocprof args...
  local name=$1 cmd=$2; shift 2
  rm -rf gpurun_out/prof_$name
  timeout 300 rocprofv3 "$@" -d gpurun_out/prof_$name -o $name -- $cmd > gpurun_out/prof_$name.log 2>&1
  echo "$name rc=$?"
}
for model in m2m splat; do
  cmd="python tools/${model}_bench.py"
  prof ${model}_stats "$cmd" --kernel-trace --stats
  python tools/rocprof_summary.py stats gpurun_out/prof_${model}_stats > gpurun_out/${TAG}_${model}_kernel_stats.txt 2>&1
  head -14 gpurun_out/${TAG}_${model}_kernel_stats.txt
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS"; do
    n=$(echo $c | tr ' ' '+' | cut -c1-24)
    prof ${model}_$n "$cmd" --kernel-trace --pmc $c
    python tools/rocprof_summary.py pmc gpurun_out/prof_${model}_$n > gpurun_out/${TAG}_${model}_pmc_$n.txt 2>&1
    head -12 gpurun_out/${TAG}_${model}_pmc_$n.txt
  done
done
rm -rf gpurun_out/prof_*/
