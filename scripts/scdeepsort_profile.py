"""Host-side profile of ScDeepSort.fit on the synthetic graph: python scripts/scdeepsort_profile.py [fp32|bf16] [n_cells] [batch]"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
from dance_amd.cellgraph import CellGeneGraph
from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
cd = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev, n_cells, n_genes, per, d = "cuda", (int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000), 2000, 200, 400
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
g = torch.Generator(device=dev).manual_seed(0)
col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32).reshape(-1)
rp_x = torch.arange(0, n_cells * per + 1, per, dtype=torch.int32, device=dev)
val_x = torch.rand(n_cells * per, device=dev, generator=g) + 0.5
rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col, val_x, n_cells, n_genes)
rowptr, gcol, gval, eid = kernels.cellgene_graph_assemble(rp_x, col, kernels.csr_row_normalize(rp_x, val_x), rp_t, col_t,
                                                          kernels.csr_row_normalize(rp_t, val_t), perm_t, n_cells, n_genes)
n_nodes = n_cells + n_genes
cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
cg = CellGeneGraph(rowptr, gcol, gval, eid, n_nodes, {"cell_id": cid, "features": torch.randn(n_nodes, d, device=dev, generator=g)})
labels = torch.randint(0, 16, (n_cells, ), generator=torch.Generator().manual_seed(0))
with tempfile.TemporaryDirectory() as tmp:
    m = ScDeepSort(d, 200, 1, "synthetic", "c3", batch_size=batch, device="cuda", save_root=tmp, verbose=False, compute_dtype=cd)
    m.fit(cg, labels, epochs=1)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    m.fit(cg, labels, epochs=1)
    torch.cuda.synchronize()
    pr.disable()
    print(cd, f"{(time.perf_counter() - t0) * 1e3:.1f} ms")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
