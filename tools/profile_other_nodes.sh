#!/bin/bash
# rocprofv3 --kernel-trace --stats of the SURVEY 8(f) nodes' device paths at 1080p (GMFSS Fortuna, IFUNet, IFRNet) + FILM: text summaries only
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
TAG=${1:?usage: profile_other_nodes.sh <tag, e.g. r06>}
for model in gmfss ifunet ifrnet film; do
  cmd="python tools/${model}_bench.py"
  [ $model = gmfss ] && cmd="$cmd --coherent"
  rm -rf gpurun_out/prof_${model}_stats
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${model}_stats -o ${model}_stats -- $cmd > gpurun_out/prof_${model}_stats.log 2>&1
  echo "$model rc=$?"
  python tools/rocprof_summary.py stats gpurun_out/prof_${model}_stats > gpurun_out/${TAG}_${model}_kernel_stats.txt 2>&1
  head -8 gpurun_out/${TAG}_${model}_kernel_stats.txt
done
rm -rf gpurun_out/prof_*/
