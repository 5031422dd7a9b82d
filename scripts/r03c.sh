#!/bin/bash
# round 3, call c: knn-k15 locality renumbering (XCD map A/B), premask default, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
DH_SPMM_XCDMAP=0 timeout 500 python scripts/locality_probe.py 1000000 all > $O/locality_map0.json 2> $O/locality_map0.err; tail -8 $O/locality_map0.err
DH_SPMM_XCDMAP=1 timeout 500 python scripts/locality_probe.py 1000000 all > $O/locality_map1.json 2> $O/locality_map1.err; tail -8 $O/locality_map1.err
timeout 300 python -m pytest tests/test_gpu_layers.py tests/test_gpu_kernels.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 300 python bench.py --no-cpu-baseline --no-x3-row > $O/bench.json 2> $O/bench.err; cut -c1-1500 $O/bench.json
