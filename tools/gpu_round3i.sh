#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
for lay in "res_c64 x32" "2440"; do
echo "== pmc $lay"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/pmcx -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/wino_bench.py "$lay" > /tmp/pmcx.log 2>&1; tail -1 /tmp/pmcx.log
python - <<PY
import csv, glob, collections
for f in glob.glob('/tmp/pmcx/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        agg[r['Kernel_Name'][:44]][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in agg.items():
        if 'wino_kernel<8' in k or 'conv_mfma2' in k:
            wc = v['SQ_WAVE_CYCLES']
            print(k, {c: f"{x / wc:.3f}" for c, x in sorted(v.items())}, f"wave_cycles {wc:.3e}")
PY
rm -rf /tmp/pmcx
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r03i.log
