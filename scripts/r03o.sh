#!/bin/bash
# round 3, call o: the long-and-narrow bf16 GEMM (tests + timing)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python scripts/gemm_bf16_rows_time.py > $O/gemm_bf16_rows.json 2> $O/rows.err; cat $O/gemm_bf16_rows.json; tail -3 $O/rows.err
