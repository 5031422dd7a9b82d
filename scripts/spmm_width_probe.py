"""SpMM rate vs row width (the feature-sliced multi-GPU mode runs width 512 / P): python scripts/spmm_width_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
dev, n, k = "cuda", 1_000_000, 15
g = torch.Generator(device=dev).manual_seed(0)
col = torch.randint(0, n, (n, k), device=dev, generator=g).sort(dim=1).values.to(torch.int32).reshape(-1)
rowptr = torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev)
val = torch.rand(n * k, device=dev, generator=g) / k
for width in (512, 256, 128, 64, 32):
    z = torch.randn(n, width, device=dev, generator=g)
    for _ in range(2):
        kernels.spmm_csr(rowptr, col, val, z, act=kernels.ACT_RELU)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        kernels.spmm_csr(rowptr, col, val, z, act=kernels.ACT_RELU)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    byt = n * k * 8.0 + 4.0 * (n + 1) + n * k * width * 4.0 + n * width * 4.0
    print(f"width {width:4d}: {dt*1e3:.3f} ms  {byt/dt/1e9:.0f} GB/s algorithmic", flush=True)

# the same 128-wide pass through the float2 configuration (8-byte aligned view): one wavefront per row, VEC = 2
big = torch.randn(n, 132, device=dev, generator=g)
z2 = big[:, 2:130]
out = torch.empty(n, 132, device=dev)[:, 2:130]
for _ in range(2):
    kernels.spmm_csr(rowptr, col, val, z2, act=kernels.ACT_RELU, out=out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    kernels.spmm_csr(rowptr, col, val, z2, act=kernels.ACT_RELU, out=out)
torch.cuda.synchronize()
print(f"width  128 via float2 lanes (G=64): {(time.perf_counter() - t) / 5 * 1e3:.3f} ms", flush=True)
