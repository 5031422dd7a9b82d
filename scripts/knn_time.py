"""Times dh_knn_bruteforce_f32 (filter) at BASELINE size and checks it against the scan on a query range:
python scripts/knn_time.py [n] [d] [k] [out.json]."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import _lib
if os.environ.get("VARIANT"):  # A/B builds: dance_amd/libdancehip_<VARIANT>.so
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libdancehip_{os.environ['VARIANT']}.so")
from dance_amd import kernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 50
k = int(sys.argv[3]) if len(sys.argv) > 3 else 15
out = sys.argv[4] if len(sys.argv) > 4 else None
g = torch.Generator(device="cuda").manual_seed(0)
res = {"n": n, "d": d, "k": k}
for name in ("randn", "clusters20"):
    x = torch.randn(n, d, device="cuda", generator=g)
    if name == "clusters20":  # the bench's knn-k15 embedding: 20 cluster centres, spread 4x the within-cluster sigma
        c = torch.randn(20, d, device="cuda", generator=g) * 4.0
        x = x + c[torch.randint(0, 20, (n,), device="cuda", generator=g)]
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    idx, dist = kernels.knn(x, k, algo=kernels.KNN_FILTER)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1)
    if ms < 2000.0:  # (a call that fell back to re-scans takes seconds: keep its single timing)
        idx, dist = kernels.knn(x, k, algo=kernels.KNN_FILTER)
        t0.record()
        for _ in range(3):
            idx, dist = kernels.knn(x, k, algo=kernels.KNN_FILTER)
        t1.record(); torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / 3
    # exactness on a query range against the scan (the scan over all 1M queries takes 3 s)
    q0, q1 = n // 2, min(n, n // 2 + 20000)
    i_s, d_s = kernels.knn(x, k, q0, q1, algo=kernels.KNN_SCAN)
    same = bool(torch.equal(i_s, idx[q0:q1]) and torch.equal(d_s, dist[q0:q1]))
    res[name] = {"filter_ms": ms, "equals_scan_on_20000_queries": same}
    print(name, ms, same, flush=True)
if out:
    json.dump(res, open(out, "w"), indent=1)
