#!/bin/bash
# round 4 development call: the repack + two-waves-per-SIMD sage kernel pair — every test shape through it, full-size parity, timing, ablations
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${TAG:-r04a}; mkdir -p $O
DANCE_AMD_SAGE_MFMA=bcm timeout 600 python -m pytest tests/test_gpu_sage_dense.py -x -q -k "mfma" > $O/sage_dense_bcm.log 2>&1; tail -5 $O/sage_dense_bcm.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "sage_mfma" > $O/fullsize.log 2>&1; tail -3 $O/fullsize.log
timeout 300 python scripts/sage_mfma_bench.py 1000000 > $O/sage_bcm.json 2> $O/sage_bcm.err; cat $O/sage_bcm.json; tail -2 $O/sage_bcm.err

[ -f $R/dance_amd/libdancehip_prof.so ] && timeout 300 python scripts/sage_prof.py > $O/prof.json 2> $O/prof.err && python - <<PY
import json
d=json.load(open("$O/prof.json"))
for k,v in d.items():
    print(k, {n:(x if not isinstance(x,list) else f"{x[0]} ({x[1]}%)") for n,x in v.items()})
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o sage -- python $R/scripts/sage_abl.py 1000000 0 > $O/abl_under_prof.json 2> $O/prof2.err
DB=$(find $O/prof -name "*.db" | head -1); python $R/scripts/rocpd_stats.py $DB $O/sage > $O/rocpd.log 2>&1; head -6 $O/rocpd.log
find $O -name "*.db" -delete
