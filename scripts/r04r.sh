#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${TAG:-r04r}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sctag_scalable.py tests/test_gpu_sctag.py tests/test_gpu_scdsc_fit.py -x -q 2>&1 | tail -3
timeout 600 python scripts/zinb_time.py > $O/zinb.json 2> $O/zinb.err; python -c "
import json; d=json.load(open('$O/zinb.json'))
for k,v in d.items(): print(k, v['ms_fwd_bwd'], v['kernels_ms'], v['loss_rel_err'], v['grad_rel_err'])"
