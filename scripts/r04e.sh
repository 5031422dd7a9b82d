#!/bin/bash
# per-kernel times of the sage kernel pair under rocprofv3 (kernel trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${TAG:-r04e}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o sage -- python $R/scripts/sage_abl.py 1000000 0 > $O/abl_under_prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1); python $R/scripts/rocpd_stats.py $DB $O/sage > $O/rocpd.log 2>&1; head -20 $O/rocpd.log
find $O -name "*.db" -delete
