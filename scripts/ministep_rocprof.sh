#!/bin/bash
# per-kernel times of the persistent steps: rocprofv3 --kernel-trace --stats over scripts/ministep_probe.py (steppers only, no fit)
# usage: scripts/ministep_rocprof.sh <tag>   (writes gpurun_out/<tag>_ministep_kernel_stats.csv)
set -u
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/msprof && mkdir -p /tmp/msprof
rocprofv3 --kernel-trace --stats -d /tmp/msprof -o ms --output-format csv -- python $ROOT/scripts/ministep_probe.py 100000 300 0 128 > /tmp/msprof/probe.json 2> /tmp/msprof/probe.err
f=$(find /tmp/msprof -name "*kernel_stats.csv" | head -1)
cp "$f" $ROOT/gpurun_out/${TAG}_ministep_kernel_stats.csv
cp /tmp/msprof/probe.json $ROOT/gpurun_out/${TAG}_ministep_probe_under_rocprof.json
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.2f} min_us {float(r['MinNs'])/1e3:8.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
