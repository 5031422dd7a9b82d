"""GraphSC.fit at 1M cells: what an epoch costs inside a multi-epoch fit, and what the final embedding read-out costs.
python scripts/graphsc_epoch_split.py [n_cells]   -> one JSON object

fit(epochs=1) = capture (if any) + one epoch + the read-out of z (1.2 GB to host numpy at 1M cells, reference graphsc.py:232-236 does the same
after the last epoch); fit(epochs=3) adds two more epochs and nothing else, so (t3 - t1) / 2 is the epoch."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
from dance_amd.cellgraph import CellGeneGraph
from dance_amd.modules.single_modality.clustering import graphsc
from dance_amd.modules.single_modality.clustering.graphsc import GraphSC

n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
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


def timed(m, epochs, bsz):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.fit(cg, epochs=epochs, batch_size=bsz)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


out = {"cells": n_cells, "genes": n_genes, "edges_per_cell": per}
for bsz, max_batch in ((8192, 2048), (8192, 8192), (128, 2048)):
    graphsc.HIPGRAPH_MAX_BATCH = max_batch
    m = GraphSC(in_feats=d, n_clusters=10, device="cuda")
    timed(m, 1, bsz)
    t1 = min(timed(m, 1, bsz) for _ in range(2))
    t3 = timed(m, 3, bsz)
    ep = (t3 - t1) / 2
    nb = -(-n_cells // bsz)
    out[f"batch={bsz} {'captured' if bsz <= max_batch else 'eager'}"] = dict(
        fit_1_epoch_s=round(t1, 4), fit_3_epochs_s=round(t3, 4), epoch_s=round(ep, 4), ms_per_batch=round(ep * 1e3 / nb, 4),
        capture_plus_readout_s=round(t1 - ep, 4), batches=nb)
print(json.dumps(out, indent=1))
