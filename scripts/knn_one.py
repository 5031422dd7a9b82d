"""One kNN call for counter collection: python scripts/knn_one.py [n] [d] [algo 0|1|2]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 50
algo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
x = torch.randn(n, d, device="cuda")
for _ in range(2):
    kernels.knn(x, 15, algo=algo)
torch.cuda.synchronize()
