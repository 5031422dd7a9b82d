#!/bin/bash
# round 3, call u: kernel trace of a ScDeepSort epoch at the reference's batch 500 (captured step), 100k cells
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03u; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/scripts/scdeepsort_profile.py fp32 100000 500 > $O/sds_500.log 2>&1
grep '^fp32' $O/sds_500.log
f=$(ls $O/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -70 "$f" > $O/sds_500_kernel_stats.csv
rm -rf $O/trace
