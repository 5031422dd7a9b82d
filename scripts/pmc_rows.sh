#!/bin/bash
# Counter passes (HBM traffic) for the kernels of the model rows that had time-only fractions until round 6: gcn_narrow_*, student_t_*, the DEC
# loss kernels (c5 row), zinb_nll_logits_* (c2 row at 100k cells) and the persistent mini-batch steps (ministep_probe).  FETCH_SIZE and WRITE_SIZE
# in their own runs (TCC counter slots), no tracing next to them.  usage: scripts/pmc_rows.sh <tag>  -> gpurun_out/<tag>_rows_pmc.json
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${TAG}_rows_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, command...
  local name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $ctr -d $OUT/${name}_$ctr -o $name --output-format csv -- "$@" > $OUT/${name}_$ctr.log 2>&1
  done
}
run c5 python $R/scripts/bench_configs.py c5_spagcn_500k_iter
run c2 python $R/scripts/bench_configs.py c2_scdsc_epoch_100k
run mini python $R/scripts/ministep_probe.py 100000 100 0 128
python $R/scripts/pmc_summary.py $(find $OUT -name "*counter_collection.csv") > $OUT/summary_all.json 2> $OUT/summary.err
python - "$OUT/summary_all.json" > $R/gpurun_out/${TAG}_rows_pmc.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
keep = ("gcn_narrow", "student_t", "dec_", "zinb", "gsc_", "sds_", "ms_grad", "gemm_small", "softmax_xent", "colsum", "axpby")
out = {"note": "mean per dispatch; FETCH_SIZE as rocprofv3 reports it (KB) and x 2 — MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-byte requests of wide coalesced "
               "reads at 64 bytes; WRITE_SIZE uncalibrated.  Counter runs serialise dispatches: mean_ms is not the kernel's time in the pipelined program.", "kernels": {}}
for k, v in d.items():
    if any(s in k for s in keep):
        e = {}
        if "FETCH_SIZE" in v:
            e["fetch_KB_reported"] = round(v["FETCH_SIZE"]["mean"], 1); e["fetch_MB_x2"] = round(v["FETCH_SIZE"]["mean"] * 2 / 1024, 3); e["dispatches"] = v["FETCH_SIZE"]["dispatches"]
        if "WRITE_SIZE" in v:
            e["write_MB_reported"] = round(v["WRITE_SIZE"]["mean"] / 1024, 3)
        out["kernels"][k] = e
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.db" -delete
tail -c 1500 $R/gpurun_out/${TAG}_rows_pmc.json
