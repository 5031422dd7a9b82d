"""GraphSC.fit's host logic on CPU tensors (mini-batch loop, block adjacency as an edge list, pos_weight / norm, the decoder
loss in its fused and dense formulations, Adam, embedding re-ordering) against tests/golden/graphsc.npz — the losses,
embeddings and weights the reference's OWN ``GraphSC.fit`` produced (graphsc.py:148-246, AST-lifted, torch-CPU over the DGL
stub; tests/golden/make_golden.py).  The HIP kernels are replaced by the torch / scipy stand-ins of tests/cpu_ops.py (the GPU
suite runs the same comparison through the kernels: tests/test_gpu_graphsc_golden.py); nothing here needs a GPU."""
import json
import os

import numpy as np
import pytest
import torch

import cpu_ops
from conftest import rel_err
from oracle import graphs as og

GOLD = os.path.join(os.path.dirname(__file__), "golden", "graphsc.npz")
STAND_INS = cpu_ops.STAND_INS


def _graph(gold):
    """The CellFeatureGraph of the golden data from the oracle's edge list (the transform itself runs HIP kernels)."""
    from dance_amd.cellgraph import CellGeneGraph
    e = og.cell_feature_graph(gold["gsc_x"], normalize_edges=False)
    n_nodes = e["n_genes"] + e["n_cells"]
    order = np.argsort(e["dst"], kind="stable")
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(e["dst"], minlength=n_nodes))
    feats = np.vstack((gold["gsc_gene_feat"], gold["gsc_cell_feat"])).astype(np.float32)
    return CellGeneGraph(torch.from_numpy(rowptr.astype(np.int32)), torch.from_numpy(e["src"][order].astype(np.int32)),
                         torch.from_numpy(e["weight"][order].astype(np.float32)), torch.from_numpy(order.astype(np.int32)), n_nodes,
                         {"cell_id": torch.from_numpy(e["cell_id"]), "feat_id": torch.from_numpy(e["feat_id"]), "features": torch.from_numpy(feats)})


@pytest.mark.parametrize("mode", ["fused", "dense"])
@pytest.mark.parametrize("tag,batch_size,agg", [("full", 64, "sum"), ("mb", 16, "sum"), ("mean", 16, "mean")])
def test_graphsc_fit_host_logic_vs_reference(monkeypatch, tag, batch_size, agg, mode):
    from dance_amd import kernels
    from dance_amd.modules.single_modality.clustering import graphsc
    for name in STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    monkeypatch.setattr(graphsc, "DECODER_MODE", mode)   # "fused": listed entries from the loader's block hook (the GPU default)
    gold = np.load(GOLD)
    kw = json.loads(str(gold["gsc_kw"]))
    kw["agg"] = agg
    m = graphsc.GraphSC(**kw, n_clusters=3, device="cpu")
    m.model.decoder.dropout = 0.0
    sd = {k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f"gsc_{tag}_sd0::")}
    m.model.load_state_dict(sd)
    m.shuffle_generator = torch.Generator().manual_seed(123)
    m.fit(_graph(gold), epochs=3, lr=1e-2, batch_size=batch_size)
    ref_losses = gold[f"gsc_{tag}_losses"]
    assert len(m.losses) == len(ref_losses)
    assert np.allclose(m.losses, ref_losses, rtol=2e-4, atol=0)
    assert rel_err(m.get_latent(), gold[f"gsc_{tag}_z"]) < 1e-3
    for k in gold.files:
        if k.startswith(f"gsc_{tag}_sd1::"):
            assert rel_err(m.model.state_dict()[k.split("::", 1)[1]].numpy(), gold[k]) < 1e-3, k
