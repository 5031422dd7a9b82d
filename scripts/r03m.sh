#!/bin/bash
# round 3, call m: kernel trace of a GraphSC epoch at 1M cells / batch 8192 (what the 4.2 ms per batch are), the AdaptiveSAGE tests after the
# exact-gather rule, ScDeepSort epochs with the capture thresholds
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_scdeepsort.py tests/test_gpu_sage_dense.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python $R/scripts/graphsc_profile.py 1000000 8192 > $O/graphsc_8192.log 2>&1 )
grep ' ms for ' $O/graphsc_8192.log
f=$(ls $O/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -45 "$f" > $O/graphsc_8192_kernel_stats.csv && head -30 $O/graphsc_8192_kernel_stats.csv | cut -c1-200
rm -rf $O/trace
timeout 600 python scripts/scdeepsort_profile.py fp32 > $O/scdeepsort_fp32.log 2>&1; grep "^fp32" $O/scdeepsort_fp32.log
