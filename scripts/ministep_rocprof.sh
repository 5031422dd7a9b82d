#!/bin/bash
# per-kernel times of the persistent steps: rocprofv3 --kernel-trace --stats over scripts/ministep_probe.py (steppers only, no fit)
# usage: scripts/ministep_rocprof.sh <tag>   (writes gpurun_out/<tag>_ministep_kernel_stats.csv)
set -u
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/msprof && mkdir -p /tmp/msprof
rocprofv3 --kernel-trace --stats -d /tmp/msprof -o ms --output-format csv -- python $ROOT/scripts/ministep_probe.py 100000 300 0 > /tmp/msprof/probe.json 2> /tmp/msprof/probe.err
f=$(find /tmp/msprof -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_ministep_kernel_stats.csv
cp /tmp/msprof/probe.json $ROOT/gpurun_out/${TAG}_ministep_probe_under_rocprof.json
head -30 "$f"
