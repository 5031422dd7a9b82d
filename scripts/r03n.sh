#!/bin/bash
# round 3, call n: GraphSC epoch split (epoch vs read-out; eager vs captured at 8192) after the small-grid split-K rule and the out-degree change;
# GEMM / GraphSC / model tests (the split-K rule changes summation order of small GEMMs)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "gemm or graphsc or scdsc or spagcn or sctag or hetero or stagate or free or narrow or linear or block" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python scripts/graphsc_epoch_split.py > $O/graphsc_epoch_split.json 2> $O/split.err; cat $O/graphsc_epoch_split.json; tail -3 $O/split.err
