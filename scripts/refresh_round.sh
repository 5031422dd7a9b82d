#!/bin/bash
# One gpurun call that regenerates everything under profiles/ for a round tag:  bash scripts/refresh_round.sh r01c
TAG=${1:-r01c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cat $OUT/bench_line.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-knn-workload > $OUT/bench_line_under_rocprof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name "*.db" | head -1); python $R/scripts/rocpd_stats.py $DB $OUT/bench > $OUT/rocpd.log 2>&1; head -12 $OUT/rocpd.log
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-knn-workload"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $OUT/sq -o sq --output-format csv -- $CMD > $OUT/sq.log 2>&1
python $R/scripts/pmc_summary.py $(find $OUT/fetch $OUT/write $OUT/sq -name "*counter_collection.csv") > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err
cd $R && python scripts/bench_rows.py > $OUT/rows.json 2> $OUT/rows.err; tail -c 600 $OUT/rows.json
# keep the merge-back small: drop the raw rocpd database, keep CSV/JSON
find $OUT -name "*.db" -delete
du -sh $OUT
