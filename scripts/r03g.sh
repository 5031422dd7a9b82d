#!/bin/bash
# round 3, call g: narrow fused layer (tests + timing + counters), (f)2 / (f)3 GPU tests, whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_narrow.py -x -q > $O/narrow_tests.log 2>&1; tail -4 $O/narrow_tests.log
timeout 300 python scripts/narrow_probe.py > $O/narrow.json 2> $O/narrow.err; tail -4 $O/narrow.err
timeout 900 python -m pytest tests/test_gpu_sctag_scalable.py tests/test_gpu_transforms.py -x -q > $O/new_tests.log 2>&1; tail -6 $O/new_tests.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
DANCE_AMD_X=1 timeout 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch --output-format csv -- python $R/scripts/narrow_probe.py > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o write --output-format csv -- python $R/scripts/narrow_probe.py > $O/write.log 2>&1
python $R/scripts/pmc_summary.py $(find $O/fetch $O/write -name "*counter_collection.csv") > $O/narrow_pmc.json 2>/dev/null
find $O -name "*.db" -delete
