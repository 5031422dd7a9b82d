#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v7; mkdir -p $O; cd $R
bash scripts/gemm_variants.sh run > $O/gemm_variants.jsonl 2> $O/gemm_variants.err; cat $O/gemm_variants.jsonl
for v in lg1 lg0; do DANCE_HIP_LIB=$R/build/variants/libdancehip_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-knn-workload --no-x3-row 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$v', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"; done
