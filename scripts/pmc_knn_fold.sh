#!/bin/bash
# counters of the fp16 kNN filter kernel (d <= 64) at n x d (default 1M x 50): bash scripts/pmc_knn_fold.sh [tag] [n] [d]
# sums over the filter launches of ONE dh_knn_bruteforce_f32 call (one per threshold pass)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_knn_fold}; mkdir -p $O
N=${2:-1000000}; D=${3:-50}
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/knn_one.py $N $D 2"
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU -d $O/a -o a --output-format csv -- $CMD > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS -d $O/b -o b --output-format csv -- $CMD > $O/b.log 2>&1
python - $(find $O -name "*counter_collection.csv") <<'PY'
import csv, sys
from collections import defaultdict
tot, calls = defaultdict(float), defaultdict(int)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        if "knn_fold_filter" not in r["Kernel_Name"]:
            continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        calls[r["Counter_Name"]] += 1
print("knn_fold_filter_kernel, per dh_knn_bruteforce_f32 call (2 calls profiled, %d filter launches each)" % (max(calls.values()) // 2 if calls else 0))
for c in sorted(tot):
    print("   %-34s %18.0f" % (c, tot[c] / 2))
PY
find $O -name "*.db" -delete
