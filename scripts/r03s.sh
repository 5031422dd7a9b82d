#!/bin/bash
# round 3, call s: kernel trace of a GraphSC epoch at the reference's batch 128 (captured step), 200k cells: what the 0.52 ms per batch consist of
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03s; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/scripts/graphsc_profile.py 200000 128 > $O/graphsc_128.log 2>&1
grep ' ms for ' $O/graphsc_128.log
f=$(ls $O/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -60 "$f" > $O/graphsc_128_kernel_stats.csv
rm -rf $O/trace
