#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v8; mkdir -p $O; cd $R
python scripts/knn_time.py > $O/knn_time.json 2> $O/knn_time.err; cat $O/knn_time.json; tail -3 $O/knn_time.err
timeout 900 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_golden_graphs.py tests/test_gpu_transforms.py -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o k --output-format csv -- python $R/scripts/knn_one.py 1000000 50 2 > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/prof/k_kernel_stats.csv")))
for r in rows[:8]: print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {r['Calls']:>4} {r['Name'][:90]}")
PY
find $O -name "*.db" -delete
