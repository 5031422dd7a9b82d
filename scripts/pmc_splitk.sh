#!/bin/bash
# memory-side counters of the gene <- cell split-K kernel at 1M cells: bash scripts/pmc_splitk.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_splitk}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/sage_splitk_time.py"
export TAG=${1:-pmc_splitk}
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum -d $O/a -o a --output-format csv -- $CMD > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/b -o b --output-format csv -- $CMD > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum -d $O/c -o c --output-format csv -- $CMD > $O/c.log 2>&1
python $R/scripts/pmc_summary.py $(find $O -name "*counter_collection.csv") 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'sage_bcm' in k:
        print(k)
        for c,x in sorted(v.items()): print('   %-34s %16d  (%.2f ms)'%(c, x['mean'], x['mean_ms']))
" | tee $O/summary.txt
tail -3 $O/c.log
find $O -name "*.db" -delete
