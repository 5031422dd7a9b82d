#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/${TAG:-r04t}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sharded_one_gpu.py -x -q 2>&1 | tail -3
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -2 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench_line.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
print("randn", d.get("x_randn",{}).get("ms_per_step"), d.get("x_randn",{}).get("kernels_ms"))
print("x3", d.get("gemm_f32x3_row",{}).get("ms_per_step"), "knn", d.get("knn_k15",{}).get("ms_per_step"))
print("cpu", d.get("cpu_baseline"))
PY
timeout 900 python bench.py --cpu-sample-cells 1000000 --steps 5 --warmup 1 --no-knn-workload --no-x3-row --no-x-randn > $O/bench_line_cpu1M.json 2> $O/bench_cpu1M.err
python - <<PY
import json
d=json.loads(open("$O/bench_line_cpu1M.json").read().strip().splitlines()[-1])
json.dump(d["cpu_baseline"], open("$O/cpu_baseline_1M.json","w"), indent=1); print(d["cpu_baseline"])
PY
