#!/bin/bash
# counter passes over the LDS-staged aggregation (separate --pmc runs, no trace flags with counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_sage
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/sage_one.py ${1:-1000000} ${2:-f32}"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS -d $OUT/sq -o sq --output-format csv -- $CMD > $OUT/sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD -d $OUT/sq2 -o sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/mem -o mem --output-format csv -- $CMD > $OUT/mem.log 2>&1
python $R/scripts/pmc_summary.py $(find $OUT -name "*counter_collection.csv") > $OUT/summary.json 2>$OUT/summary.err
python - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    print(k)
    for c,x in v.items(): print("   %-28s %.5g  (%.3f ms, n=%d)"%(c,x["mean"],x["mean_ms"],x["dispatches"]))
PY
tail -3 $OUT/sq2.log
