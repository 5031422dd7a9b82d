#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v1; mkdir -p $O; cd $R
bash scripts/gemm_variants.sh run > $O/gemm_variants.jsonl 2> $O/gemm_variants.err; cat $O/gemm_variants.jsonl
python scripts/gemm_x3_bench.py > $O/x3.json 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py tests/test_gpu_fullsize.py tests/test_gpu_gemm_x3.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python $R/scripts/gemm_only.py > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python $R/scripts/gemm_only.py > $O/write.log 2>&1
python $R/scripts/pmc_summary.py $(find $O/fetch $O/write -name "*counter_collection.csv") > $O/pmc_gemm.json 2> $O/pmc.err; cat $O/pmc_gemm.json | head -40
find $O -name "*.db" -delete
