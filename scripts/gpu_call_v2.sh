#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v2; mkdir -p $O; cd $R
bash scripts/gemm_variants.sh run > $O/gemm_variants.jsonl 2> $O/gemm_variants.err; cat $O/gemm_variants.jsonl
python scripts/gemm_x3_bench.py > $O/x3.json 2>&1; cat $O/x3.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/rb -o rb --output-format csv -- python $R/scripts/rocblas_name.py > $O/rb.log 2>&1
grep -h "Cijk\|gemm" $O/rb/*kernel_stats.csv | cut -c1-400 | head -5
find $O -name "*.db" -delete
