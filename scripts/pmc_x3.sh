#!/bin/bash
# counters of gemm_f32x3_kernel at the headline shapes (one pass: SQ counters only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/px3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS -d /tmp/px3 -o x3 --output-format csv -- python $R/scripts/gemm_x3_time.py > /tmp/px3.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/px3b -o x3 --output-format csv -- python $R/scripts/gemm_x3_time.py > /tmp/px3b.log 2>&1
python $R/scripts/pmc_summary.py $(find /tmp/px3 /tmp/px3b -name "*counter_collection.csv") 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'x3' in k and 'reduce' not in k:
        print(k[:80]); [print('  ',c, int(x['mean']), x['dispatches'], round(x['mean_ms'],3),'ms') for c,x in sorted(v.items())]
"
