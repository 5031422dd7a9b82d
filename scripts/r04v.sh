#!/bin/bash
# GPU: kernel-level split of the gene <- cell call (rocprofv3 --kernel-trace --stats)
TAG=${TAG:-r04v}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/$TAG/prof -o splitk --output-format csv -- python scripts/sage_splitk_time.py > gpurun_out/$TAG/run.log 2>&1
tail -3 gpurun_out/$TAG/run.log
f=$(find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-220
cp "$f" gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/prof
