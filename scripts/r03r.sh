#!/bin/bash
# round 3, call r: HBM bytes of the forward SpMM on knn-k15, cells in random order vs renumbered by locality (FETCH_SIZE / WRITE_SIZE, separate passes)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03r; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for ord in none rcm; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c -d $O/${ord}_$c -o p --output-format csv -- python $R/scripts/knn_spmm_traffic.py $ord > $O/${ord}_$c.log 2>&1
  done
  python $R/scripts/pmc_summary.py $(find $O/${ord}_FETCH_SIZE $O/${ord}_WRITE_SIZE -name "*counter_collection.csv") > $O/pmc_$ord.json 2> $O/pmc_$ord.err
  python - <<PY
import json
d=json.load(open("$O/pmc_$ord.json"))
for k,v in d.items():
    if "spmm_slice128" in k: print("$ord", k[:60], {c:(round(x["mean"]),round(x["mean_ms"],3)) for c,x in v.items()})
PY
done
find $O -name "*.db" -delete; du -sh $O
