#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v6; mkdir -p $O; cd $R
python scripts/gemm_variants.py default > $O/gemm.json 2> $O/gemm.err; cat $O/gemm.json
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py tests/test_gpu_fullsize.py tests/test_gpu_gemm_x3.py -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -5
