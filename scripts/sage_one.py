"""One LDS-staged cell <- gene aggregation at n cells (for rocprofv3 passes): python scripts/sage_one.py [n_cells] [f32|bf16]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

dev = "cuda"
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
n_genes, dfeat, per = 2000, 400, 200
g = torch.Generator(device=dev).manual_seed(0)
col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32)
col = torch.cat((col, (n_genes + torch.arange(n_cells, device=dev, dtype=torch.int32))[:, None]), 1).reshape(-1).contiguous()
rowptr = torch.arange(0, n_cells * (per + 1) + 1, per + 1, dtype=torch.int32, device=dev)
w = torch.rand(col.numel(), device=dev, generator=g) + 0.5
feats = torch.randn(n_genes + n_cells, dfeat, device=dev, generator=g)
if dt == "bf16":
    feats = feats.to(torch.bfloat16)
cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
cid_cells = cid[n_genes:].contiguous()
alpha = torch.rand(n_genes + 2, device=dev, generator=g) + 0.5
for _ in range(3):
    kernels.sage_aggregate_cells(rowptr, col, w, cid, cid_cells, alpha, feats, 0, n_genes)
torch.cuda.synchronize()
