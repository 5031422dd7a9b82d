#!/bin/bash
# Round-6 refresh (one gpurun call): everything refresh_round.sh collects for the tag + the persistent steps' probe and per-kernel listing,
# the counter passes of the model-row kernels, the reference-batch epochs.   TAG=r06final bash scripts/refresh_r06.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
TAG=${TAG:-r06final}
bash scripts/refresh_round.sh $TAG
O=$R/gpurun_out/$TAG
timeout 600 python scripts/ministep_probe.py 100000 500 1 > $O/ministep_probe.json 2> $O/ministep_probe.err; tail -c 400 $O/ministep_probe.json
timeout 600 python scripts/ministep_probe.py 1000000 500 0 128 > $O/ministep_probe_1M.json 2> $O/ministep_probe_1M.err
bash scripts/ministep_rocprof.sh $TAG > $O/ministep_rocprof.log 2>&1; mv $R/gpurun_out/${TAG}_ministep_kernel_stats.csv $O/ 2>/dev/null; mv $R/gpurun_out/${TAG}_ministep_probe_under_rocprof.json $O/ 2>/dev/null
timeout 600 python scripts/ref_batch_epochs.py 100000 > $O/ref_batch_epochs_100k.json 2> $O/ref_100k.err
bash scripts/pmc_rows.sh $TAG > $O/pmc_rows.log 2>&1; mv $R/gpurun_out/${TAG}_rows_pmc.json $O/rows_pmc.json 2>/dev/null; rm -rf $R/gpurun_out/${TAG}_rows_pmc
find $O -name "*.db" -delete
du -sh $O
