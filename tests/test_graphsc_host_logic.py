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


def test_fit_full_graph_equals_the_reference_fit_with_one_batch(monkeypatch):
    """GraphSC.fit_full_graph (the cell-sharded full-graph form, here on one process: same arithmetic, no collectives) == the
    reference's own fit with batch_size >= n_cells (golden tag "full"): losses, embedding, updated parameters."""
    from dance_amd import kernels
    from dance_amd.modules.single_modality.clustering import graphsc
    for name in STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    gold = np.load(GOLD)
    kw = json.loads(str(gold["gsc_kw"]))
    m = graphsc.GraphSC(**kw, n_clusters=3, device="cpu")
    m.model.decoder.dropout = 0.0
    m.model.load_state_dict({k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("gsc_full_sd0::")})
    m.ops = cpu_ops
    m.fit_full_graph(_graph(gold), epochs=3, lr=1e-2)
    assert np.allclose(m.losses, gold["gsc_full_losses"], rtol=2e-4, atol=0)
    assert rel_err(m.get_latent(), gold["gsc_full_z"]) < 1e-3
    for k in gold.files:
        if k.startswith("gsc_full_sd1::"):
            assert rel_err(m.model.state_dict()[k.split("::", 1)[1]].numpy(), gold[k]) < 1e-3, k


def test_static_cell_block_matches_sampled_block(monkeypatch):
    """cellgraph.StaticCellBlock (sources = [seeds | all genes], e_max entries, one padding row) gives WeightedGraphConv / GCNAE the
    outputs and gradients of the dgl.to_block-ordered block of the same seeds; the captured step's body (_CapturedStep._step, run
    eagerly here) performs the same update as one eager batch of GraphSC.fit."""
    from dance_amd import kernels
    from dance_amd.cellgraph import MultiLayerFullNeighborSampler, StaticCellBlock
    from dance_amd.modules.single_modality.clustering import graphsc
    for name in STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    gold = np.load(GOLD)
    g = _graph(gold)
    g.ndata["order"] = g.ndata["label"] = g.ndata["feat_id"]
    n_genes = g.gene_prefix()
    kw = json.loads(str(gold["gsc_kw"]))
    torch.manual_seed(0)
    seeds = torch.tensor([n_genes + i for i in (5, 0, 17, 9, 30, 2)])
    _, _, blocks = MultiLayerFullNeighborSampler(1).sample(g, seeds, True)
    sb = StaticCellBlock(g, seeds.numel())
    sb.seeds.copy_(seeds)
    sb.rebuild()
    assert int(sb.bad) == 0 and sb.rowptr.numel() == seeds.numel() + 2 and int(sb.rowptr[-1]) == sb.e_max
    assert torch.equal(sb.in_degrees(), blocks[0].in_degrees())
    for agg in ("sum", "mean"):
        m = graphsc.GCNAE(**{**kw, "agg": agg})
        outs = []
        for blk in (blocks[0], sb):
            for p in m.parameters():
                p.grad = None
            _, z = m.forward([blk], blk.srcdata["features"], decode=False)
            z.sum().backward()
            outs.append((z.detach().clone(), [p.grad.clone() for p in m.parameters()]))
        assert rel_err(outs[1][0].numpy(), outs[0][0].numpy()) < 1e-6
        for a, b in zip(outs[1][1], outs[0][1]):
            assert rel_err(a.numpy(), b.numpy()) < 1e-5
    # a non-cell seed is flagged
    sb.seeds[0] = 0
    sb.rebuild()
    assert int(sb.bad) & 1   # bit 1: not a cell of the layout (bit 2, "no single self loop", is set as well)
    # the step body == one eager batch (dropout off): same loss, same updated parameters
    fit = graphsc.GraphSC(**kw, n_clusters=3, device="cpu")
    fit.model.dropout = None
    fit.model.decoder.dropout = 0.0
    ref = graphsc.GraphSC(**kw, n_clusters=3, device="cpu")
    ref.model.load_state_dict(fit.model.state_dict())
    ref.model.dropout, ref.model.decoder.dropout = None, 0.0
    opt = torch.optim.Adam(fit.model.parameters(), lr=1e-2)
    step = graphsc._CapturedStep(fit, g, 6, opt)
    step.block.seeds.copy_(seeds)
    emb, loss = step._step()
    ropt = torch.optim.Adam(ref.model.parameters(), lr=1e-2)
    blk = blocks[0]
    _, e1 = ref.model.forward([blk], blk.srcdata["features"], decode=False)
    b = 6.0
    us = torch.arange(6, dtype=torch.int32)
    rl = (b * b / ((b * b - b) * 2)) * graphsc.gram_listed_bce(ref.model.forward([blk], blk.srcdata["features"], decode=False)[1], us, us, (b * b - b) / b)
    ropt.zero_grad()
    rl.backward()
    ropt.step()
    assert abs(float(loss) - float(rl)) < 1e-6 * abs(float(rl)) and rel_err(emb.numpy(), e1.detach().numpy()) < 1e-6
    for (k, a), (_, c) in zip(fit.model.state_dict().items(), ref.model.state_dict().items()):
        assert rel_err(a.numpy(), c.numpy()) < 1e-5, k


def check_weighted_graph_conv_alpha(device):
    """WeightedGraphConvAlpha (graphsc.py:487-566) against the reference's OWN class run over the DGL graph stub on torch-CPU
    (tests/golden/wgc_alpha.npz): forward and the gradients of the features, the external weight and alpha — norm "both", tanh, a bias;
    node ids as [N, 1] tensors (what the reference's np.where needs) and alpha as the [G + 2, 1] parameter of scDeepSort."""
    import types
    from dance_amd.modules.single_modality.clustering.graphsc import WeightedGraphConvAlpha
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "wgc_alpha.npz"))
    n_genes, n_cells, d, h = (int(v) for v in gold["wa_dims"])
    n = n_genes + n_cells
    dense, ids = gold["wa_dense"], gold["wa_ids"]
    rowptr = np.concatenate([[0], np.cumsum(dense.sum(1))]).astype(np.int32)
    col = np.concatenate([np.flatnonzero(r) for r in dense]).astype(np.int32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    blk = types.SimpleNamespace(rowptr=t(rowptr), col=t(col), srcdata={"id": t(ids)[:, None]}, dstdata={"id": t(ids)[:, None]},
                                number_of_dst_nodes=lambda: n, number_of_src_nodes=lambda: n, has_zero_in_degree=lambda: False)
    feat, w, alpha = (t(gold[k]).requires_grad_(True) for k in ("wa_feat", "wa_w", "wa_alpha"))
    layer = WeightedGraphConvAlpha(d, h, norm="both", weight=False, bias=True, activation=torch.tanh).to(device)
    with torch.no_grad():
        layer.bias.copy_(t(gold["wa_bias"]))
    out = layer(blk, feat, weight=w, alpha=alpha, gene_num=n_genes)
    out.backward(t(gold["wa_dy"]))
    assert rel_err(out.detach().cpu().numpy(), gold["wa_out"]) < 1e-5
    for got, key in ((feat.grad, "wa_dfeat"), (w.grad, "wa_dw"), (alpha.grad, "wa_dalpha")):
        assert got.shape == gold[key].shape and rel_err(got.cpu().numpy(), gold[key]) < 1e-5, key
    with pytest.raises(RuntimeError):   # the module's own weight and an external one together (:534-539)
        WeightedGraphConvAlpha(d, h).to(device)(blk, feat, weight=w, alpha=alpha, gene_num=n_genes)


def test_weighted_graph_conv_alpha_vs_reference(monkeypatch):
    from dance_amd import kernels
    for name in cpu_ops.STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    check_weighted_graph_conv_alpha("cpu")
