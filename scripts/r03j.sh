#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
timeout 300 python scripts/narrow_probe.py > $O/narrow.json 2> $O/narrow.err; tail -5 $O/narrow.err
QUICK=1 bash scripts/refresh_round.sh r03j
