"""GraphConvolution 50 -> 50 fwd+bwd (BASELINE config 5's layer) at 1M rows, k = 15: fused narrow kernels vs the generic chain.
   python scripts/narrow_probe.py > gpurun_out/narrow.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import autograd, kernels  # noqa: E402
from dance_amd.autograd import gcn_layer  # noqa: E402
from dance_amd.graph import CSRGraph  # noqa: E402

dev = "cuda"
n, k, f = 1_000_000, 15, 50
g = torch.Generator(device=dev).manual_seed(0)
colk = torch.randint(0, n, (n, k), device=dev, generator=g).sort(dim=1).values.to(torch.int32).reshape(-1)
graph = CSRGraph(torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev), colk, torch.full((n * k, ), 1 / 15., device=dev), n, n)
graph.transpose()
w = (torch.randn(f, f, device=dev, generator=g) / 7).requires_grad_(True)
b = torch.zeros(f, device=dev, requires_grad=True)
dy = torch.randn(n, f, device=dev, generator=g)
x50 = torch.randn(n, f, device=dev, generator=g)
x52 = torch.zeros(n, 52, device=dev)
x52[:, :f] = x50
x64 = torch.zeros(n, 64, device=dev)
x64[:, :f] = x50
out = {}
alg_bytes = 1.85e9  # B_min accounting of the 50 -> 50 layer fwd + bwd
gather_bytes = n * k * (8 + f * 4) + 4 * n + n * f * 4 + 2 * n * 64 * 4 + n * f * 4  # fwd: edges + gathered rows + Y + agg; bwd: agg + dY
for label, x, fused in (("fused, X ld=50 (8-byte rows, 2 rows/wave)", x50, True), ("fused, X ld=52 (16-byte rows, 4 rows/wave)", x52[:, :f], True),
                        ("fused, X ld=64 (rows = two aligned 128-byte lines)", x64[:, :f], True), ("generic chain (round 2)", x50, False)):
    autograd.NARROW_FUSED = fused

    def step():
        w.grad = b.grad = None
        gcn_layer(x, w, graph, b, False).backward(dy)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with kernels.KernelTimer() as t:
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
    ksum = {kname: round(v[1], 4) for kname, v in sorted(t.summary().items())}
    out[label] = {"ms_fwd_bwd": ms, "kernels_ms": ksum, "kernel_sum_ms": round(sum(ksum.values()), 4),
                  "hbm_frac_of_8TBs_on_kernel_sum": round(alg_bytes / (sum(ksum.values()) * 1e-3) / 8e12, 4),
                  "hbm_frac_gather_accounting": round(gather_bytes / (sum(ksum.values()) * 1e-3) / 8e12, 4)}
    print(label, out[label], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
