#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v10; mkdir -p $O; cd $R
python scripts/knn_time.py > $O/knn_time.json 2> $O/knn_time.err; cat $O/knn_time.json
timeout 900 python -m pytest tests/test_gpu_graphs.py -m gpu -x -q > $O/tests.log 2>&1; grep -E "passed|failed|rror" $O/tests.log | tail -5
bash scripts/pmc_knn2.sh v10/pmc 2>&1 | grep -E "filter|MFMA_BUSY|WAIT_ANY|WAVE_CYCLES|INSTS_VALU |INSTS_BRANCH|INSTS_SALU|GUI"
