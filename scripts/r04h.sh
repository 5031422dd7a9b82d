#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${TAG:-r04h}; mkdir -p $O
timeout 300 python scripts/sage_abl.py 1000000 0,16,32,64,80,20,22,86 > $O/abl.json 2> $O/abl.err; cat $O/abl.json
DANCE_AMD_SM2_ABL=32 DANCE_AMD_SAGE_MFMA=bcm timeout 600 python -m pytest tests/test_gpu_sage_dense.py -x -q -k "mfma" 2>&1 | tail -2
