#!/bin/bash
# SQ counter passes over one kNN call (separate --pmc runs, no trace flags)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_knn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/knn_one.py 300000 50"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVES -d $OUT/sq1 -o sq1 --output-format csv -- $CMD > $OUT/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $OUT/sq2 -o sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ -d $OUT/sq3 -o sq3 --output-format csv -- $CMD > $OUT/sq3.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LEVEL_WAVES SQC_DCACHE_BUSY_CYCLES -d $OUT/sq4 -o sq4 --output-format csv -- $CMD > $OUT/sq4.log 2>&1
python $R/scripts/pmc_summary.py $OUT/*/*/*counter_collection.csv > $OUT/summary.json 2>$OUT/summary.err
cat $OUT/summary.json | head -150
grep -il "error\|invalid" $OUT/*.log
