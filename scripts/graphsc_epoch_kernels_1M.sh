#!/bin/bash
# every kernel of one GraphSC.fit epoch at 1M cells, batch 8192 (two rocprofv3 kernel traces, 1 and 3 epochs, differenced)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for e in 1 3; do
  rm -rf /tmp/gk$e
  rocprofv3 --kernel-trace --stats -d /tmp/gk$e -o gk --output-format csv -- python $R/scripts/epoch_kernels.py run graphsc $e 1000000 8192 > /tmp/gk$e.log 2>&1
done
a=$(find /tmp/gk1 -name "*kernel_stats.csv" | head -1); b=$(find /tmp/gk3 -name "*kernel_stats.csv" | head -1)
python $R/scripts/epoch_kernels.py diff $a $b 2 > $R/gpurun_out/${TAG:-r06}_graphsc_epoch_kernels_1M_b8192.md
head -30 $R/gpurun_out/${TAG:-r06}_graphsc_epoch_kernels_1M_b8192.md
