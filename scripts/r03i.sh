#!/bin/bash
# round 3, call i: narrow v3 timing, captured GraphSC step (tests + epochs at batch 128 / 8192, 100k and 1M cells)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_narrow.py tests/test_gpu_graphsc_golden.py tests/test_gpu_block.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python scripts/narrow_probe.py > $O/narrow.json 2> $O/narrow.err; tail -4 $O/narrow.err
timeout 600 python scripts/ref_batch_epochs.py 100000 > $O/ref_batch_epochs_100k.json 2> $O/ref_100k.err; tail -5 $O/ref_100k.err
DANCE_AMD_HIPGRAPH=0 timeout 600 python scripts/ref_batch_epochs.py 100000 > $O/ref_batch_epochs_100k_eager.json 2> $O/ref_100k_eager.err; tail -5 $O/ref_100k_eager.err
timeout 900 python scripts/ref_batch_epochs.py 1000000 graphsc > $O/ref_batch_epochs_1M.json 2> $O/ref_1M.err; tail -5 $O/ref_1M.err
