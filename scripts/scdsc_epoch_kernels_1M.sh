#!/bin/bash
# every kernel of one ScDSC.fit epoch at 1M cells (two rocprofv3 kernel traces, 2 and 6 epochs, differenced)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for e in 2 6; do
  rm -rf /tmp/ek$e
  rocprofv3 --kernel-trace --stats -d /tmp/ek$e -o ek --output-format csv -- python $R/scripts/epoch_kernels.py run scdsc $e 1000000 > /tmp/ek$e.log 2>&1
done
a=$(find /tmp/ek2 -name "*kernel_stats.csv" | head -1); b=$(find /tmp/ek6 -name "*kernel_stats.csv" | head -1)
python $R/scripts/epoch_kernels.py diff $a $b 4 > $R/gpurun_out/${TAG:-r06}_scdsc_epoch_kernels_1M.md
head -45 $R/gpurun_out/${TAG:-r06}_scdsc_epoch_kernels_1M.md
