"""Three launches of the two headline GEMMs (S = X W, dW = X^T dS) for profiling runs; `rocblas` as argv[1] runs torch.mm instead."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
M = 1_000_000
X = torch.randn(M, 2000, device=dev)
W = torch.randn(2000, 512, device=dev) / 45
D = torch.randn(M, 512, device=dev)
rocblas = len(sys.argv) > 1 and sys.argv[1] == "rocblas"
for _ in range(3):
    if rocblas:
        torch.mm(X, W)
        torch.mm(X.t(), D)
    else:
        kernels.gemm(X, W, mode="exact")
        kernels.gemm(X, D, trans_a=True, mode="exact")
torch.cuda.synchronize()
