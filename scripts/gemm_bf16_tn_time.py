"""GPU: dW = dY^T X on the bf16 matrix cores at the C3 shape (1M cells, 200 x 400) and its HBM rate.  gpurun_out/$TAG/tn.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

dev = "cuda"
out = {}
for n, m, k in ((200, 400, 1_000_000), (200, 400, 65536), (16, 200, 1_000_000), (224, 512, 1_000_000)):
    dy = torch.randn(k, n, device=dev).to(torch.bfloat16)
    x = torch.randn(k, m, device=dev).to(torch.bfloat16)
    fn = lambda: kernels.gemm_bf16(dy, x, trans_a=True, out_dtype=torch.float32)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    byt = k * (n + m) * 2.0
    ref = dy.double().t() @ x.double()
    err = float((fn().double() - ref).abs().max() / ref.abs().max())
    out[f"dW {n}x{m} K={k}"] = dict(ms=ms, hbm_GBs=byt / ms / 1e6, frac_hbm=byt / ms / 1e6 / 8000, tflops=2.0 * n * m * k / ms / 1e9, err=err)
    print(n, m, k, out[f"dW {n}x{m} K={k}"], flush=True)
tag = os.path.join("gpurun_out", os.environ.get("TAG", "tn"))
os.makedirs(tag, exist_ok=True)
json.dump(out, open(os.path.join(tag, "tn.json"), "w"), indent=1)
