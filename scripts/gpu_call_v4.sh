#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v5; mkdir -p $O; cd $R
python scripts/spmm_mask_probe.py > $O/spmm_probe.json 2> $O/spmm_probe.err; cat $O/spmm_probe.json; tail -2 $O/spmm_probe.err
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py tests/test_gpu_fullsize.py tests/test_gpu_rccl.py tests/test_gpu_models.py -m gpu -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; cat $O/bench.json
