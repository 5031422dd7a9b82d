#!/usr/bin/env python
"""The forward SpMM of the headline layer on knn-k15 at 1M cells in ONE node order (argv[1] = none | rcm), three launches — run under
`rocprofv3 --pmc FETCH_SIZE` to see how much of the gathered operand still comes from HBM (scripts/r03r.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dance_amd import kernels  # noqa: E402
from dance_amd.graph import CSRGraph, locality_order  # noqa: E402

order = sys.argv[1] if len(sys.argv) > 1 else "none"
n, H, K = 1_000_000, bench.N_HIDDEN, bench.K_NEIGH
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(7)
centers = torch.randn((20, 50), device=dev, generator=g) * 4.0
label = torch.randint(0, 20, (n, ), device=dev, generator=g)
emb = centers[label] + torch.randn((n, 50), device=dev, generator=g)
idx, dist_ = kernels.knn(emb, K)
(rowptr, col, val), _ = kernels.umap_connectivities(idx, dist_.contiguous())
graph = CSRGraph(rowptr, col, val, n, n, symmetric=True)
if order == "rcm":
    graph = graph.permute(locality_order(graph).to(dev))
z = torch.randn((n, H), device=dev, generator=torch.Generator(device=dev).manual_seed(11))
y = torch.empty_like(z)
mask = torch.empty(kernels.relu_mask_bytes(n, H), dtype=torch.uint8, device=dev)
for _ in range(3):
    kernels.spmm_csr_relu(graph.rowptr, graph.col, graph.val, z, n_cols=n, act=kernels.ACT_RELU, out_mask=mask, out=y)
torch.cuda.synchronize()
print(order, "nnz", graph.nnz)
