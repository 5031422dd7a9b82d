#!/bin/bash
# PMC passes for the headline bench (run through gpurun).  Counters are collected in their own runs, separate
# from the kernel-trace/--stats run, one counter family per pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_r01
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $OUT/sq -o sq --output-format csv -- $CMD > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -d $OUT/l2 -o l2 --output-format csv -- $CMD > $OUT/l2.log 2>&1
ls -R $OUT | head -40
tail -3 $OUT/*.log
