"""profiles/hbm_traffic.json from a PMC summary (scripts/pmc_summary.py output):
   python scripts/make_hbm_traffic.py gpurun_out/r01c/pmc_summary.json > profiles/hbm_traffic.json
HBM bytes per launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE
reports half the bytes of wide coalesced reads — MI355X_MICROARCH.md §HBM, calibrated in round 1 on relu_backward)."""
import json
import sys

d = json.load(open(sys.argv[1]))
names = {  # bench.py kernel key -> (device kernel name, launches per C-ABI call)
    "gemm_f32_nn": ("gemm_f32_kernel<Cfg<2, 4, 4, 2>, false, false, true>", 1),
    "gemm_f32_tn": ("gemm_f32_kernel<Cfg<2, 4, 4, 2>, true, false, true>", 1),
    # the 512-wide SpMM runs as four 128-column passes (spmm.hip): bytes per call = 4 x bytes per launch
    "spmm_csr_f32[fwd]": ("spmm_slice128_kernel<false, true, true>", 4),   # records the ReLU bitmap
    "spmm_csr_f32[bwd]": ("spmm_slice128_kernel<false, false, true>", 4),  # plain gather of the pre-masked dY (round 3 default)
    "relu_mask_apply_f32": ("relu_mask_apply_kernel", 1),                  # dY * [Y > 0] from the bitmap, one streaming pass
}
out = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, "
                   "scripts/refresh_round.sh, bench.py at 1M cells); bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024: on "
                   "gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md HBM section; "
                   "confirmed in round 1 on relu_backward: 2.05 GB raw for 4.10 GB read). See profiles/README.md."}
raw = {}
for k, v in d.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        f, w = v["FETCH_SIZE"]["mean"] * 1024, v["WRITE_SIZE"]["mean"] * 1024
        raw[k] = {"FETCH_SIZE_bytes_raw": f, "WRITE_SIZE_bytes": w, "hbm_bytes_corrected": 2 * f + w,
                  "ms": v["FETCH_SIZE"]["mean_ms"]}
for key, (sub, launches) in names.items():
    if sub in raw:
        out[key] = raw[sub]["hbm_bytes_corrected"] * launches
# the kernel sources these counters belong to: bench.py refuses to report them for any other build
import hashlib
import os
_csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dance_amd", "csrc")
out["source_sha256"] = {f: hashlib.sha256(open(os.path.join(_csrc, f), "rb").read()).hexdigest() for f in ("gemm_f32.hip", "spmm.hip", "common.h")}
out["raw"] = raw
print(json.dumps(out, indent=1))
