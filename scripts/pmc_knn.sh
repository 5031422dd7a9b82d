#!/bin/bash
# kernel trace + counter passes over one kNN call (separate --pmc runs, no trace flags with counters)
#   bash scripts/pmc_knn.sh [n] [d] [algo]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-400000}; D=${2:-50}; ALGO=${3:-2}
OUT=$R/gpurun_out/pmc_knn
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/knn_one.py $N $D $ALGO"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
DB=$(find $OUT/trace -name "*.db" | head -1); python $R/scripts/rocpd_stats.py $DB $OUT/knn > $OUT/rocpd.log 2>&1; head -12 $OUT/rocpd.log
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/l2 -o l2 --output-format csv -- $CMD > $OUT/l2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $OUT/sq -o sq --output-format csv -- $CMD > $OUT/sq.log 2>&1
python $R/scripts/pmc_summary.py $(find $OUT/fetch $OUT/l2 $OUT/sq -name "*counter_collection.csv") > $OUT/summary.json 2>$OUT/summary.err
python - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    if "knn" in k:
        print(k)
        for c,x in v.items(): print("   %-28s %.4g  (%.2f ms, n=%d)"%(c,x["mean"],x["mean_ms"],x["dispatches"]))
PY
find $OUT -name "*.db" -delete
