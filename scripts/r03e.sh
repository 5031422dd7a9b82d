#!/bin/bash
# round 3, call e: new GPU tests (free riders, dh_comm world 1, config-4 full size, pruned decoder), emulated ranks, reference batch sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_free_riders.py tests/test_gpu_rccl.py tests/test_gpu_graphsc_golden.py tests/test_gpu_sage_dense.py "tests/test_gpu_fullsize.py::test_graphsc_batch_full_size" -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python scripts/emulate_rank.py > $O/emulate_rank.json 2> $O/emulate_rank.err; tail -8 $O/emulate_rank.err
timeout 900 python scripts/ref_batch_epochs.py 100000 > $O/ref_batch_epochs.json 2> $O/ref_batch_epochs.err; tail -6 $O/ref_batch_epochs.err
