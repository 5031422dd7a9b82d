#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${TAG:-r04m}; mkdir -p $O
timeout 300 python scripts/sage_prof.py > $O/timeline.txt 2> $O/timeline.err; cat $O/timeline.txt; tail -3 $O/timeline.err
DANCE_AMD_SAGE_MFMA=bcm timeout 600 python -m pytest tests/test_gpu_sage_dense.py -x -q -k "mfma" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "sage_mfma" 2>&1 | tail -2
timeout 300 python scripts/sage_mfma_bench.py 1000000 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print({k:(v['mfma_path_ms'],v['rel_diff_vs_gather']) for k,v in d.items()})"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o sage -- python $R/scripts/sage_abl.py 1000000 0 > $O/abl_under_prof.json 2> $O/prof2.err
DB=$(find $O/prof -name "*.db" | head -1); python $R/scripts/rocpd_stats.py $DB $O/sage > $O/rocpd.log 2>&1; head -6 $O/rocpd.log | cut -c1-150
find $O -name "*.db" -delete
