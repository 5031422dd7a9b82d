#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/gemm_only.py"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD -d $OUT/sq2 -o sq2 --output-format csv -- $CMD > $OUT/sq2.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum -d $OUT/tcp -o tcp --output-format csv -- $CMD > $OUT/tcp.log 2>&1
timeout 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $OUT/tcp2 -o tcp2 --output-format csv -- $CMD > $OUT/tcp2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/sq3 -o sq3 --output-format csv -- $CMD > $OUT/sq3.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum -d $OUT/ta -o ta --output-format csv -- $CMD > $OUT/ta.log 2>&1
ls $OUT/*/ | head -30
grep -il "error\|invalid" $OUT/*.log
