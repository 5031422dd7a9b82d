import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels
dev = torch.device("cuda:0")
M = 1_000_000
X = torch.randn(M, 2000, device=dev); W = torch.randn(2000, 512, device=dev) / 45; D = torch.randn(M, 512, device=dev)
for _ in range(3):
    kernels.gemm(X, W, mode="exact"); kernels.gemm(X, D, trans_a=True, mode="exact")
torch.cuda.synchronize()
