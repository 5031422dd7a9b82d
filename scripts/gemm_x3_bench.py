"""Time dh_gemm_f32x3 against dh_gemm_f32 at the headline layer's two GEMM shapes (HIP events on torch's current stream)."""
import json
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X = torch.randn(n, 2000, device="cuda")
W = torch.randn(2000, 512, device="cuda")
dZ = torch.randn(n, 512, device="cuda")
out = {}
for name, fn in {
        "nn x3": lambda: kernels.gemm(X, W, mode="x3"),
        "nn exact": lambda: kernels.gemm(X, W, mode="exact"),
        "tn x3": lambda: kernels.gemm(X, dZ, trans_a=True, mode="x3"),
        "tn exact": lambda: kernels.gemm(X, dZ, trans_a=True, mode="exact"),
}.items():
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    out[name] = {"ms": round(ms, 3), "TFLOPs_fp32_equiv": round(2 * n * 2000 * 512 / ms / 1e9, 1)}
print(json.dumps(out, indent=1))
