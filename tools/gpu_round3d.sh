#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
for ab in 0 15; do
echo "== pmc lds, ablate $ab"
VFI_WINO_ABLATE=$ab timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL --kernel-trace -d /tmp/pmc$ab -o p --output-format csv -- python $GRAFT_REPO_ROOT/tools/wino_bench.py "res_c64 x32" > /tmp/pmc$ab.log 2>&1; tail -2 /tmp/pmc$ab.log
python - <<PY
import csv, glob, collections
for f in glob.glob('/tmp/pmc$ab/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        agg[r['Kernel_Name'][:44]][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in agg.items():
        if 'wino' in k or 'conv_mfma2' in k:
            print(k, {c: f"{x:.3e}" for c, x in sorted(v.items())})
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r03d.log
