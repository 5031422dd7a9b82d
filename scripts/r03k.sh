#!/bin/bash
# round 3, call k: ScDeepSort captured step (test + epochs at batch 500), the round's rows (bench_rows.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_gpu_scdeepsort.py tests/test_gpu_bf16.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 600 python scripts/ref_batch_epochs.py 100000 > $O/ref_batch_epochs_100k.json 2> $O/ref_100k.err; tail -5 $O/ref_100k.err
timeout 1500 python scripts/bench_rows.py > $O/rows.json 2> $O/rows.err; tail -3 $O/rows.err; tail -c 1500 $O/rows.json
