"""bf16 GEMM rate probe: python scripts/gemm_bf16_probe.py  (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
for M, N, K, tb in [(1_000_000, 512, 2048, True), (8192, 8192, 8192, True), (1_000_000, 200, 400, True), (4096, 4096, 4096, True), (100_000, 2048, 2048, True)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(2):
        kernels.gemm_bf16(a, b, trans_b=tb)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        kernels.gemm_bf16(a, b, trans_b=tb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print(M, N, K, f"{dt*1e3:.3f} ms", f"{2*M*N*K/dt/1e12:.0f} TFLOP/s", flush=True)
