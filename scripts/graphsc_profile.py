"""Host-side profile of GraphSC.fit on a synthetic cell-gene graph: python scripts/graphsc_profile.py [n_cells] [batch]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
from dance_amd.cellgraph import CellGeneGraph
from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev, n_genes, per, d = "cuda", 2000, 200, 50
g = torch.Generator(device=dev).manual_seed(0)
col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32).reshape(-1)
rp_x = torch.arange(0, n_cells * per + 1, per, dtype=torch.int32, device=dev)
val_x = torch.rand(n_cells * per, device=dev, generator=g) + 0.5
rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col, val_x, n_cells, n_genes)
rowptr, gcol, gval, eid = kernels.cellgene_graph_assemble(rp_x, col, val_x, rp_t, col_t, val_t, perm_t, n_cells, n_genes)
n_nodes = n_cells + n_genes
cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
fid = torch.cat((-torch.ones(n_genes, dtype=torch.int32), torch.arange(n_cells, dtype=torch.int32))).to(dev)
cg = CellGeneGraph(rowptr, gcol, gval, eid, n_nodes, {"cell_id": cid, "feat_id": fid, "features": torch.randn(n_nodes, d, device=dev, generator=g)})
m = GraphSC(in_feats=d, n_clusters=10, device="cuda")
m.fit(cg, epochs=1, batch_size=bsz)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
m.fit(cg, epochs=1, batch_size=bsz)
torch.cuda.synchronize()
pr.disable()
print(f"{(time.perf_counter() - t0) * 1e3:.1f} ms for {n_cells} cells, batch {bsz}")
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
pstats.Stats(pr).sort_stats("cumtime").print_stats(70)
