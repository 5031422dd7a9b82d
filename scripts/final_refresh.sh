#!/bin/bash
# final refresh of a round (TAG=r03v bash scripts/final_refresh.sh): everything refresh_round.sh collects + the epoch measurements at the reference's batch sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash scripts/refresh_round.sh ${TAG:-r03v}
O=$R/gpurun_out/${TAG:-r03v}
timeout 600 python scripts/ref_batch_epochs.py 100000 > $O/ref_batch_epochs_100k.json 2> $O/ref_100k.err; tail -c 600 $O/ref_batch_epochs_100k.json
timeout 600 python scripts/graphsc_epoch_split.py > $O/graphsc_epoch_split.json 2> $O/split.err; tail -c 900 $O/graphsc_epoch_split.json
