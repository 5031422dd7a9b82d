#!/usr/bin/env python
"""Column-slice width of the headline aggregation: the 128-column passes gather from a 512 MB working set of which the 256 MB
MALL holds about half; 64-column passes gather from 256 MB.  Times Y = relu(A S) as width / w passes of the generic kernel
(dh_spmm_csr_f32 on column views; w = 128 also through the fused slice kernel) on rand-k15 and checks the result bit for bit.
    python scripts/spmm_slice_width_probe.py > gpurun_out/spmm_slice_width_probe.json"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dance_amd import kernels  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
H, K = bench.N_HIDDEN, bench.K_NEIGH
rowptr, col, val = bench.synth_rand_graph(n, K, dev, seed=1)
s = torch.randn((n, H), device=dev)
y = torch.empty((n, H), device=dev)
mask = torch.empty(kernels.relu_mask_bytes(n, H), dtype=torch.uint8, device=dev)
out = {}


def t_ms(fn, it=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / it * 1e3, 3)


ref = kernels.spmm_csr_relu(rowptr, col, val, s, n_cols=n, act=kernels.ACT_RELU, out_mask=mask).clone()
out["fused 128-column slices"] = t_ms(lambda: kernels.spmm_csr_relu(rowptr, col, val, s, n_cols=n, act=kernels.ACT_RELU, out_mask=mask, out=y))
for w in (512, 256, 128, 64, 32, 16):
    def run():
        for c in range(0, H, w):
            kernels.spmm_csr(rowptr, col, val, s[:, c:c + w], n_cols=n, act=kernels.ACT_RELU, out=y[:, c:c + w])
    y.zero_()
    out[f"generic, {H // w} passes of {w} columns"] = t_ms(run)
    out[f"generic {w}: identical"] = bool(torch.equal(y, ref))
print(json.dumps(out, indent=1))
