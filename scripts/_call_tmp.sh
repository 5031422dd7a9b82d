R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v14; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_sage_dense.py -m gpu -x -q -k mfma > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 300 python scripts/sage_mfma_bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err
