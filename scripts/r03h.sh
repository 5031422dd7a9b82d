#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_narrow.py "tests/test_gpu_transforms.py::test_dense_to_csr_matches_numpy_nonzero" -x -q > $O/narrow_tests.log 2>&1; tail -3 $O/narrow_tests.log
timeout 300 python scripts/narrow_probe.py > $O/narrow.json 2> $O/narrow.err; tail -4 $O/narrow.err
timeout 300 python scripts/hipgraph_probe.py 128 > $O/hipgraph_128.json 2> $O/hipgraph_128.err; cat $O/hipgraph_128.json; tail -3 $O/hipgraph_128.err
timeout 300 python scripts/hipgraph_probe.py 8192 > $O/hipgraph_8192.json 2> $O/hipgraph_8192.err; cat $O/hipgraph_8192.json
