"""A few launches of dh_gram_sigmoid_f32 at B x d (default 8192 x 300) for counter collection: python scripts/gram_one.py [B] [d] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
d = int(sys.argv[2]) if len(sys.argv) > 2 else 300
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
z = torch.randn(B, d, device="cuda") * 0.1
for _ in range(reps):
    kernels.gram_sigmoid(z)
torch.cuda.synchronize()
