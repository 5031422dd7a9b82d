"""Times dh_gemm_f32 (NN: S = X W, TN: dW = X^T dS) of the library DANCE_HIP_LIB points at, at the headline shapes,
and checks sampled outputs against float64.  One JSON line per call; driven by scripts/gemm_variants.sh."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "default"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(M, 2000, device=dev, generator=g)
W = torch.randn(2000, 512, device=dev, generator=g) / 45
D = torch.randn(M, 512, device=dev, generator=g)


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


flops = 2 * M * 2000 * 512
S = kernels.gemm(X, W, mode="exact")
rows = torch.randint(0, M, (256,), device=dev, generator=g)
ref = X[rows].double() @ W.double()
err_nn = float((S[rows].double() - ref).abs().max() / ref.abs().max())
dW = kernels.gemm(X, D, trans_a=True, mode="exact")
cols = torch.arange(0, 2000, 97, device=dev)
ref = X[:, cols].double().t() @ D.double()
err_tn = float((dW[cols].double() - ref).abs().max() / ref.abs().max())
t_nn = timed(lambda: kernels.gemm(X, W, mode="exact"))
t_tn = timed(lambda: kernels.gemm(X, D, trans_a=True, mode="exact"))
print(json.dumps({"variant": name, "M": M, "nn_ms": round(t_nn, 4), "nn_TF": round(flops / t_nn / 1e9, 2), "tn_ms": round(t_tn, 4),
                  "tn_TF": round(flops / t_tn / 1e9, 2), "err_nn": err_nn, "err_tn": err_tn}), flush=True)
if name == "p3m1":  # rocBLAS on the same box, once
    t_nn, t_tn = timed(lambda: torch.mm(X, W)), timed(lambda: torch.mm(X.t(), D))
    print(json.dumps({"variant": "rocblas(torch.mm)", "M": M, "nn_ms": round(t_nn, 4), "nn_TF": round(flops / t_nn / 1e9, 2),
                      "tn_ms": round(t_tn, 4), "tn_TF": round(flops / t_tn / 1e9, 2)}), flush=True)
