"""Scratch probe (not part of the product): time each kernel of the GCN path at the headline size and
calibrate against rocBLAS (torch.mm) on the same box."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from dance_amd import kernels  # noqa: E402
from dance_amd.graph import CSRGraph  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
F, H, k = 2000, 512, 15


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters


g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(N, F, device=dev, generator=g)
W = torch.randn(F, H, device=dev, generator=g) / 45
col = torch.randint(0, N, (N, k), device=dev, generator=g, dtype=torch.int64).sort(dim=1).values.to(torch.int32).reshape(-1)
rowptr = torch.arange(0, N * k + 1, k, device=dev, dtype=torch.int32)
val = torch.full((N * k,), 1.0 / k, device=dev)
graph = CSRGraph(rowptr, col, val, N, N)
t0 = time.time(); gt = graph.transpose(); torch.cuda.synchronize(); t_tr = time.time() - t0
dY = torch.randn(N, H, device=dev, generator=g)
res = {"N": N, "transpose_s": t_tr}
Z = kernels.gemm(X, W)
res["gemm_nn_ms"] = timeit(lambda: kernels.gemm(X, W))
res["gemm_nn_tflops"] = 2 * N * F * H / res["gemm_nn_ms"] / 1e9
res["rocblas_nn_ms"] = timeit(lambda: torch.mm(X, W))
res["gemm_nn_maxdiff_vs_rocblas"] = float((Z - torch.mm(X, W)).abs().max())
Y = kernels.spmm_csr(graph.rowptr, graph.col, graph.val, Z, act=1)
res["spmm_fwd_ms"] = timeit(lambda: kernels.spmm_csr(graph.rowptr, graph.col, graph.val, Z, act=1))
bytes_spmm = N * k * 8 + 4 * (N + 1) + N * k * H * 4 + N * H * 4
res["spmm_fwd_GBs"] = bytes_spmm / res["spmm_fwd_ms"] / 1e6
G = kernels.relu_backward(Y, dY)
res["relu_bwd_ms"] = timeit(lambda: kernels.relu_backward(Y, dY))
dS = kernels.spmm_csr(gt.rowptr, gt.col, gt.val, G)
res["spmm_bwd_ms"] = timeit(lambda: kernels.spmm_csr(gt.rowptr, gt.col, gt.val, G))
res["spmm_bwd_GBs"] = bytes_spmm / res["spmm_bwd_ms"] / 1e6
dW = kernels.gemm(X, dS, trans_a=True)
res["gemm_tn_ms"] = timeit(lambda: kernels.gemm(X, dS, trans_a=True))
res["gemm_tn_tflops"] = 2 * N * F * H / res["gemm_tn_ms"] / 1e9
res["rocblas_tn_ms"] = timeit(lambda: torch.mm(X.t(), dS))
res["gemm_tn_reldiff_vs_rocblas"] = float((dW - torch.mm(X.t(), dS)).abs().max() / dW.abs().max())
res["gemm_nt_ms"] = timeit(lambda: kernels.gemm(dS, W, trans_b=True), iters=3, warm=1)
res["rocblas_nt_ms"] = timeit(lambda: torch.mm(dS, W.t()), iters=3, warm=1)
# HBM copy calibration
a = torch.empty(1 << 30, device=dev, dtype=torch.uint8); b = torch.empty_like(a)
res["copy_GBs"] = 2 * a.numel() / timeit(lambda: b.copy_(a)) / 1e6
res["total_ms"] = res["gemm_nn_ms"] + res["spmm_fwd_ms"] + res["relu_bwd_ms"] + res["spmm_bwd_ms"] + res["gemm_tn_ms"]
res["cells_per_s"] = N / res["total_ms"] * 1e3
print(json.dumps(res, indent=1))
