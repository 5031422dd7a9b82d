"""dh_gemm_bf16 at a compute-heavy shape and at the C3 dense-update shapes, plus the big-d kNN filter path: ms / TFLOP/s."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402


def timed(fn, it=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


out = {}
bf = torch.bfloat16
for name, (M, K, N) in {"NT 1M x 2048 x 512": (1_000_000, 2048, 512), "NT 8192^3": (8192, 8192, 8192), "NT 1M x 400 -> 200 (C3 fwd)": (1_000_000, 400, 200)}.items():
    A = torch.randn(M, K, device="cuda").to(bf)
    B = torch.randn(N, K, device="cuda").to(bf)
    ms = timed(lambda: kernels.gemm_bf16(A, B, trans_b=True))
    out[name] = {"ms": round(ms, 3), "TFLOPs": round(2 * M * K * N / ms / 1e9, 1)}
    del A, B
x = torch.randn(100_000, 2000, device="cuda")
kernels.knn(x, 15, algo=2)
torch.cuda.synchronize()
t0 = time.perf_counter()
kernels.knn(x, 15, algo=2)
torch.cuda.synchronize()
out["knn filter 100k x 2000"] = {"ms": round((time.perf_counter() - t0) * 1e3, 1)}
print(json.dumps(out, indent=1))
