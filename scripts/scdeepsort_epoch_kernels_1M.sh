#!/bin/bash
# every kernel of one ScDeepSort.fit epoch at 1M cells, bf16, batch 65536 (two rocprofv3 kernel traces, 1 and 7 epochs, differenced)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for e in 1 7; do
  rm -rf /tmp/sk$e
  rocprofv3 --kernel-trace --stats -d /tmp/sk$e -o sk --output-format csv -- python $R/scripts/epoch_kernels.py run scdeepsort $e > /tmp/sk$e.log 2>&1
done
a=$(find /tmp/sk1 -name "*kernel_stats.csv" | head -1); b=$(find /tmp/sk7 -name "*kernel_stats.csv" | head -1)
python $R/scripts/epoch_kernels.py diff $a $b 6 > $R/gpurun_out/${TAG:-r06}_scdeepsort_epoch_kernels_1M.md
head -60 $R/gpurun_out/${TAG:-r06}_scdeepsort_epoch_kernels_1M.md
