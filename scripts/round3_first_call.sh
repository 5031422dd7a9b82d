#!/bin/bash
# First gpurun call of the next round: everything that was prepared at the end of round 2 without GPU time left.
#   bash scripts/round3_first_call.sh [tag]      (about 2 GPU-minutes)
# 1. the gated tests of dh_gram_listed_* and the GraphSC decoder mode that uses them
# 2. GraphSC epoch in every decoder mode (400k cells) + the decoder kernel against its unfused form
# 3. backward-SpMM mask A/B at the headline shape
# 4. counters of dh_gram_sigmoid_f32 (matrix-core busy, waits, LDS conflicts): why 0.63 of the fp32 peak
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r03a}; mkdir -p $O
cd $R
DANCE_AMD_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_gpu_graphsc_golden.py -q > $O/experimental_tests.log 2>&1; tail -3 $O/experimental_tests.log
DANCE_AMD_EXPERIMENTAL=1 timeout 120 python scripts/graphsc_decoder_time.py 400000 > $O/graphsc_decoder.json 2> $O/graphsc_decoder.err; tail -5 $O/graphsc_decoder.err
timeout 60 python scripts/bwd_mask_ab.py > $O/bwd_mask_ab.json 2>/dev/null; cat $O/bwd_mask_ab.json
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/gram_one.py 8192 300 3"
timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/a -o a --output-format csv -- $CMD > $O/a.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD -d $O/b -o b --output-format csv -- $CMD > $O/b.log 2>&1
python $R/scripts/pmc_summary.py $(find $O/a $O/b -name "*counter_collection.csv") > $O/gram_pmc.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$O/gram_pmc.json"))
for k, v in d.items():
    if "gram_sigmoid" in k:
        print(k)
        for c, x in sorted(v.items()):
            print("   %-30s %16d" % (c, x["mean"]))
PY
find $O -name "*.db" -delete
