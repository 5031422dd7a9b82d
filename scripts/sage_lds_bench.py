"""1M-cell timing of the LDS-staged cell <- gene aggregation vs the generic gather kernel (fp32 and bf16)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

dev = "cuda"
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_genes, dfeat, per = 2000, 400, 200
g = torch.Generator(device=dev).manual_seed(0)
col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32)
col = torch.cat((col, (n_genes + torch.arange(n_cells, device=dev, dtype=torch.int32))[:, None]), 1).reshape(-1).contiguous()
rowptr = torch.arange(0, n_cells * (per + 1) + 1, per + 1, dtype=torch.int32, device=dev)
w = torch.rand(col.numel(), device=dev, generator=g) + 0.5
feats = torch.randn(n_genes + n_cells, dfeat, device=dev, generator=g)
cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
cid_cells = cid[n_genes:].contiguous()
alpha = torch.rand(n_genes + 2, device=dev, generator=g) + 0.5


def ms(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


e = col.numel()
out = {}
for name, h in (("f32", feats), ("bf16", feats.to(torch.bfloat16))):
    s = 4 if name == "f32" else 2
    byt = e * 8.0 + 4.0 * (n_cells + 1) + n_genes * dfeat * s + n_cells * dfeat * s
    old = kernels.sage_aggregate if name == "f32" else kernels.sage_aggregate_bf16
    t_old = ms(lambda: old(rowptr, col, w, cid, cid_cells, alpha, h))
    _, ws = kernels.sage_aggregate_cells(rowptr, col, w, cid, cid_cells, alpha, h, 0, n_genes)
    t_new = ms(lambda: kernels.sage_aggregate_cells(rowptr, col, w, cid, cid_cells, alpha, h, 0, n_genes))
    t_reuse = ms(lambda: kernels.sage_aggregate_cells(rowptr, col, w, cid, cid_cells, alpha, h, 0, n_genes, workspace=ws, reuse_segments=True))
    a = old(rowptr, col, w, cid, cid_cells, alpha, h).float()
    b = kernels.sage_aggregate_cells(rowptr, col, w, cid, cid_cells, alpha, h, 0, n_genes)[0].float()
    t_dense = ms(lambda: kernels.sage_aggregate_dense(rowptr, col, w, cid, cid_cells, alpha, h, 0, n_genes))
    c = kernels.sage_aggregate_dense(rowptr, col, w, cid, cid_cells, alpha, h, 0, n_genes).float()
    out[name] = dict(dense_ms=t_dense, hbm_frac_dense=byt / t_dense / 1e6 / 8000, dense_max_abs_diff=float((a - c).abs().max()), gather_ms=t_old, lds_ms=t_new, lds_reuse_segments_ms=t_reuse, algorithmic_GB=byt / 1e9,
                     hbm_frac_gather=byt / t_old / 1e6 / 8000, hbm_frac_lds=byt / t_reuse / 1e6 / 8000,
                     lds_delivery_TBs=e * dfeat * 4.0 / t_reuse / 1e9, max_abs_diff=float((a - b).abs().max()), ref_max=float(a.abs().max()))
# gene <- cell rows of the same graph (the transposed direction): gather (one wavefront per 1e5-edge row) vs dense
rp_t, col_t, val_t, _ = kernels.csr_transpose(rowptr, col, w, n_cells, n_genes + n_cells)
rp_g = rp_t[:n_genes + 1].contiguous()           # rows = genes; columns = cell ROW indices of the cell-row CSR
e_g = int(rp_g[-1])
col_g = (col_t[:e_g] + n_genes).contiguous()     # as node ids: cells are nodes [G, G+N)
val_g = val_t[:e_g].contiguous()
cid_genes = cid[:n_genes].contiguous()
for name, h in (("f32", feats), ("bf16", feats.to(torch.bfloat16))):
    old = kernels.sage_aggregate if name == "f32" else kernels.sage_aggregate_bf16
    t_old = ms(lambda: old(rp_g, col_g, val_g, cid, cid_genes, alpha, h), iters=2)
    mx = int((rp_g[1:] - rp_g[:-1]).max())
    t_dense = ms(lambda: kernels.sage_aggregate_dense(rp_g, col_g, val_g, cid, cid_genes, alpha, h, n_genes, n_cells, dst_are_genes=True, max_row_nnz=mx), iters=2)
    a = old(rp_g, col_g, val_g, cid, cid_genes, alpha, h).float()
    c = kernels.sage_aggregate_dense(rp_g, col_g, val_g, cid, cid_genes, alpha, h, n_genes, n_cells, dst_are_genes=True, max_row_nnz=mx).float()
    out["gene_rows_" + name] = dict(gather_ms=t_old, dense_ms=t_dense, edges=e_g, max_abs_diff=float((a - c).abs().max()), ref_max=float(a.abs().max()))
print(json.dumps(out, indent=1))
