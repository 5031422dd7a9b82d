#!/bin/bash
# development: time of the graph-sc step with the forward / decoder kernels cut after phase n (DANCE_AMD_MINISTEP_DBG=f,d)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for dbg in 0,0 1,0 2,0 0,1 0,2; do
  echo -n "dbg=$dbg: "
  DANCE_AMD_MINISTEP_DBG=$dbg python $ROOT/scripts/ministep_probe.py 100000 500 0 128 2>&1 | grep "^128" | cut -c1-60
done
