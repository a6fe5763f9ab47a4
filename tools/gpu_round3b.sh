#!/bin/bash
# round 3: where does the Winograd kernel's time go?  ablations (timing only) + counters on the block-3 ResConv shape
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for ab in 0 1 2 3 4 7 8 15; do echo "== ablate $ab"; VFI_WINO_ABLATE=$ab timeout 120 python tools/wino_bench.py "res_c64 x32" "2440" 2>&1 | grep "rife\|film"; done
cd /tmp
echo "== pmc"; timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --kernel-trace -d /tmp/pmc1 -o p1 --output-format csv -- python $GRAFT_REPO_ROOT/tools/wino_bench.py "res_c64 x32" > /tmp/pmc1.log 2>&1; tail -3 /tmp/pmc1.log
python - <<'PY'
import csv, glob, collections
for f in glob.glob('/tmp/pmc1/**/*counter_collection.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r['Kernel_Name'][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in agg.items():
        if 'wino' in k or 'conv_mfma2' in k:
            print(k, {c: f"{x:.3e}" for c, x in v.items()})
PY
} 2>&1 | tee $GRAFT_REPO_ROOT/gpurun_out/r03b.log | tail -80
