"""Fused vs unfused inner-product decoder of graph-sc at the batch shape of the C4 row (B = 8192, d = 300), and one
GraphSC.fit epoch in both modes: python scripts/graphsc_decoder_time.py [n_cells]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
from dance_amd.cellgraph import CellGeneGraph
from dance_amd.modules.single_modality.clustering import graphsc

dev = "cuda"
out = {}


def gpu_ms(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


B, d = 8192, 300
z = torch.randn(B, d, device=dev) * 0.1
scale = torch.tensor([1e-8], device=dev)


def dense():
    x = kernels.gemm(z, z, trans_b=True)
    r = kernels.softplus_rowsum(x)
    dx = kernels.sigmoid_scale(x, scale)
    return r, kernels.gemm(dx, z), kernels.gemm(dx, z, trans_a=True)


ms_f = gpu_ms(lambda: kernels.gram_sigmoid(z))
ms_d = gpu_ms(dense)
fl = 4.0 * B * B * 320
out["gram_sigmoid B=8192 d=300"] = dict(ms=ms_f, TFLOPs=fl / ms_f / 1e9, frac_of_f32_mfma_peak=fl / ms_f / 1e9 / 157.3)
out["unfused (z z^T, softplus pass, sigmoid pass, two dZ GEMMs)"] = dict(ms=ms_d)
r, o = kernels.gram_sigmoid(z)
x = z.double() @ z.double().t()
out["check"] = dict(rowloss_rel=float((r.double() - torch.nn.functional.softplus(x).sum(1)).abs().max() / torch.nn.functional.softplus(x).sum(1).abs().max()),
                    o_rel=float((o.double() - torch.sigmoid(x) @ z.double()).abs().max() / (torch.sigmoid(x) @ z.double()).abs().max()))
del x
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
n_genes, per, din = 2000, 200, 50
g = torch.Generator(device=dev).manual_seed(0)
col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32).reshape(-1)
rp_x = torch.arange(0, n_cells * per + 1, per, dtype=torch.int32, device=dev)
val_x = torch.rand(n_cells * per, device=dev, generator=g) + 0.5
rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col, val_x, n_cells, n_genes)
rowptr, gcol, gval, eid = kernels.cellgene_graph_assemble(rp_x, col, val_x, rp_t, col_t, val_t, perm_t, n_cells, n_genes)
n_nodes = n_cells + n_genes
cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
fid = torch.cat((-torch.ones(n_genes, dtype=torch.int32), torch.arange(n_cells, dtype=torch.int32))).to(dev)
cg = CellGeneGraph(rowptr, gcol, gval, eid, n_nodes, {"cell_id": cid, "feat_id": fid, "features": torch.randn(n_nodes, din, device=dev, generator=g)})
for mode in ("fused", "dense"):
    graphsc.DECODER_MODE = mode
    torch.manual_seed(0)
    m = graphsc.GraphSC(in_feats=din, n_clusters=10, device="cuda")
    m.shuffle_generator = torch.Generator().manual_seed(1)
    m.fit(cg, epochs=1, batch_size=B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.fit(cg, epochs=1, batch_size=B)
    torch.cuda.synchronize()
    out[f"GraphSC.fit epoch cells={n_cells} batch={B} decoder={mode}"] = dict(ms=(time.perf_counter() - t0) * 1e3, last_losses=m.losses[-3:])
    print(mode, out[f"GraphSC.fit epoch cells={n_cells} batch={B} decoder={mode}"], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
