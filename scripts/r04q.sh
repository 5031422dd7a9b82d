#!/bin/bash
# full GPU suite + ZINB timing
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${TAG:-r04q}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python scripts/zinb_time.py > $O/zinb.json 2> $O/zinb.err; cat $O/zinb.json; tail -2 $O/zinb.err
