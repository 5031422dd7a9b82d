#!/bin/bash
# One gpurun call that regenerates everything under profiles/ for a round tag:  bash scripts/refresh_round.sh r01c
# QUICK=1 skips the per-row benchmarks and the 1M-cell CPU baseline.
# NOPMC=1 skips the counter passes (profiles/hbm_traffic.json stays as committed: valid while gemm_f32.hip / spmm.hip / common.h are unchanged),
#         the 1M-cell CPU baseline and the GEMM / kNN counter comparisons.
TAG=${1:-r01c}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-knn-workload --no-x3-row --no-x-randn --no-configs > $OUT/bench_line_under_rocprof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name "*.db" | head -1); python $R/scripts/rocpd_stats.py $DB $OUT/bench > $OUT/rocpd.log 2>&1; head -12 $OUT/rocpd.log
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-knn-workload --no-x3-row --no-x-randn --no-configs"
if [ -z "$NOPMC" ]; then
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d $OUT/sq -o sq --output-format csv -- $CMD > $OUT/sq.log 2>&1
python $R/scripts/pmc_summary.py $(find $OUT/fetch $OUT/write $OUT/sq -name "*counter_collection.csv") > $OUT/pmc_summary.json 2> $OUT/pmc_summary.err
python $R/scripts/make_hbm_traffic.py $OUT/pmc_summary.json > $OUT/hbm_traffic.json 2> $OUT/hbm_traffic.err
# second bench line with the traffic file of THIS build in place (roofline.traffic is only reported for matching sources)
cp $OUT/hbm_traffic.json $R/profiles/hbm_traffic.json
fi
cd $R; python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; cat $OUT/bench_line.json; cd /tmp
[ -n "$QUICK$NOPMC" ] || ( cd $R && timeout 400 python bench.py --cpu-sample-cells 1000000 --steps 5 --warmup 1 --no-knn-workload --no-x3-row --no-x-randn --no-configs > $OUT/bench_line_cpu1M.json 2> $OUT/bench_cpu1M.err )
[ -n "$QUICK" ] || ( cd $R && python scripts/bench_rows.py > $OUT/rows.json 2> $OUT/rows.err; tail -c 600 $OUT/rows.json )
[ -n "$QUICK$NOPMC" ] || ( bash $R/scripts/pmc_gemm2.sh $TAG/pmc_gemm2 > $OUT/gemm_pmc_vs_rocblas.json 2> $OUT/gemm_pmc.err; bash $R/scripts/pmc_knn2.sh $TAG/pmc_knn2 > $OUT/knn_filter_pmc.txt 2>&1 )
# keep the merge-back small: drop the raw rocpd database, keep CSV/JSON
find $OUT -name "*.db" -delete
du -sh $OUT
