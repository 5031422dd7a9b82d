#!/bin/bash
# MFMA-busy / clock / wait counters of our two GEMMs and of rocBLAS on the same shapes (separate --pmc passes, no tracing).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-pmc_gemm2}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for who in ours rocblas; do
  arg=""; [ $who = rocblas ] && arg=rocblas
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/a_$who -o a --output-format csv -- python $R/scripts/gemm_only.py $arg > $O/a_$who.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $O/b_$who -o b --output-format csv -- python $R/scripts/gemm_only.py $arg > $O/b_$who.log 2>&1
done
python - <<PY
import csv, glob, json, re, collections
out = {}
for who in ("ours", "rocblas"):
    acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for path in glob.glob("$O/*_%s/**/*counter_collection.csv" % who, recursive=True):
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if not ("gemm_f32_kernel" in name or name.startswith("Cijk")):
                continue
            short = ("TN" if ("true, false" in name or "Ailk_Bljk" in name) else "NN") if "gemm_f32_kernel" in name or "Cijk" in name else name
            a = acc[(short, r["Counter_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for (k, c), (n, v, ms) in sorted(acc.items()):
        out.setdefault(who + " " + k, {})[c] = [round(v / n), round(ms / n, 3)]
print(json.dumps(out, indent=1))
PY
find $O -name "*.db" -delete
