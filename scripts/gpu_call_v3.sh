#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v3; mkdir -p $O; cd $R
bash scripts/gemm_variants.sh run > $O/gemm_variants.jsonl 2> $O/gemm_variants.err; cat $O/gemm_variants.jsonl; tail -3 $O/gemm_variants.err
