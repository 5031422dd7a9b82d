#!/bin/bash
# counters of the sage kernel pair at 1M x 2000 x 10 %, D = 400: bash scripts/pmc_sage.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_sage}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/sage_abl.py 1000000 0"
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/a -o a --output-format csv -- $CMD > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/b -o b --output-format csv -- $CMD > $O/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_LDS -d $O/c -o c --output-format csv -- $CMD > $O/c.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum -d $O/d -o d --output-format csv -- $CMD > $O/d.log 2>&1
python $R/scripts/pmc_summary.py $(find $O -name "*counter_collection.csv") 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'sage_bcm_kernel' in k:
        print(k)
        for c,x in sorted(v.items()): print('   %-34s %16d  (%.2f ms)'%(c, x['mean'], x['mean_ms']))
"
tail -3 $O/d.log
find $O -name "*.db" -delete
