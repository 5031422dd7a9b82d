"""Wall time of the filter-path kNN (k = 15) at a few sizes + identity with the scan at 100k: python scripts/knn_time.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

out = {}
g = torch.Generator(device="cuda").manual_seed(0)
for n, d in ((100_000, 50), (1_000_000, 50), (1_000_000, 32), (400_000, 64)):
    centers = torch.randn(20, d, device="cuda", generator=g) * 4
    x = centers[torch.randint(0, 20, (n,), device="cuda", generator=g)] + torch.randn(n, d, device="cuda", generator=g)
    kernels.knn(x, 15, algo=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx, dist = kernels.knn(x, 15, algo=2)
    torch.cuda.synchronize()
    out[f"filter n={n} d={d}"] = round((time.perf_counter() - t0) * 1e3, 2)
    if n == 100_000:
        i2, d2 = kernels.knn(x, 15, algo=1)
        out["filter == scan (100k)"] = bool(torch.equal(idx, i2) and torch.equal(dist, d2))
print(json.dumps(out, indent=1))
