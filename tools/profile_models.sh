#!/bin/bash
# rocprofv3 evidence for the M2M and FILM paths at 1080p: kernel stats + FETCH_SIZE / WRITE_SIZE passes (separate runs)
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-r01b}
prof() {  # name, command, rocprof args...
  local name=$1 cmd=$2; shift 2
  rm -rf gpurun_out/prof_$name
  timeout 240 rocprofv3 "$@" -d gpurun_out/prof_$name -o $name -- $cmd > gpurun_out/prof_$name.log 2>&1
  echo "$name rc=$?"
}
for model in m2m film; do
  cmd="python tools/${model}_bench.py"
  prof ${model}_stats "$cmd" --kernel-trace --stats
  python tools/rocprof_summary.py stats gpurun_out/prof_${model}_stats > gpurun_out/${TAG}_${model}_kernel_stats.txt 2>&1
  head -14 gpurun_out/${TAG}_${model}_kernel_stats.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    prof ${model}_$c "$cmd" --kernel-trace --pmc $c
    python tools/rocprof_summary.py pmc gpurun_out/prof_${model}_$c > gpurun_out/${TAG}_${model}_pmc_$c.txt 2>&1
    head -8 gpurun_out/${TAG}_${model}_pmc_$c.txt
  done
done
rm -rf gpurun_out/prof_*/
