#!/bin/bash
# round 3, call f: the new GPU tests ((f)2 scalable scTAG + ZINB, (f)3 device pipeline + filters), then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_sctag_scalable.py tests/test_gpu_transforms.py tests/test_gpu_sctag.py tests/test_gpu_scdsc_fit.py tests/test_gpu_scheteronet.py -x -q > $O/new_tests.log 2>&1; tail -6 $O/new_tests.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
