"""GPU: AdaptiveSAGE / GNN / ScDeepSort (BASELINE config 1: scDeepSort plumbing on a small synthetic subset)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import rel_err
from oracle import graphs as og
from oracle import sage as osg

pytestmark = pytest.mark.gpu


def _graph(n_cells, n_genes, d, seed, dev):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import CellFeatureGraph
    rng = np.random.default_rng(seed)
    x = ((rng.random((n_cells, n_genes)) < 0.2) * rng.integers(1, 9, (n_cells, n_genes))).astype(np.float32)
    data = Data(AnnDataLite(x, obsm={"f": rng.standard_normal((n_cells, d)).astype(np.float32)},
                            varm={"f": rng.standard_normal((n_genes, d)).astype(np.float32)}))
    CellFeatureGraph("f")(data)
    return x, data.data.uns["CellFeatureGraph"]


def test_adaptive_sage_block_matches_oracle(cuda_device):
    from dance_amd.cellgraph import NeighborSampler
    from dance_amd.nn import AdaptiveSAGE
    n_cells, n_genes, d, hid = 90, 50, 40, 24
    x, g = _graph(n_cells, n_genes, d, 0, cuda_device)
    ref = og.cell_feature_graph(x)
    seeds = torch.tensor([n_genes + 5, n_genes + 17, n_genes + 3, n_genes + 80], device=cuda_device)
    _, out_nodes, blocks = NeighborSampler([-1]).sample(g, seeds)
    blk = blocks[0]
    assert blk.number_of_dst_nodes() == 4 and torch.equal(blk.srcdata["_ID"][:4], seeds)
    alpha = nn.Parameter(torch.rand(n_genes + 2, 1, device=cuda_device) + 0.5)
    layer = AdaptiveSAGE(d, hid, alpha, nn.Identity(), nn.ReLU(), nn.Identity()).to(cuda_device)
    h = blk.srcdata["features"]
    z = layer(blk, h)
    # (1) the discarded aggregation == oracle restatement of gnn.py:62-82,90 on the block's edges
    src_ids = blk.srcdata["_ID"].cpu().numpy()
    lut = -np.ones(n_cells + n_genes, dtype=np.int64)
    lut[src_ids] = np.arange(src_ids.size)
    sel = np.isin(ref["dst"], seeds.cpu().numpy())
    seed_pos = {int(s): i for i, s in enumerate(seeds.cpu().numpy())}
    e_src, e_dst = lut[ref["src"][sel]], np.array([seed_pos[int(v)] for v in ref["dst"][sel]])
    cid = ref["cell_id"][src_ids]
    neigh_ref = osg.sage_neigh(e_src, e_dst, ref["weight"][sel], cid, cid[:4], alpha.detach().cpu().numpy(), h.cpu().numpy(), 4)
    assert rel_err(layer.last_neigh.cpu().numpy(), neigh_ref) < 1e-5
    # (2) the layer output follows gnn.py:92-96: act(Linear(h_dst)), neigh unused, alpha gets no gradient
    lin = layer.layers[1]
    z_ref = torch.relu(h[:4].cpu() @ lin.weight.detach().cpu().t() + lin.bias.detach().cpu())
    assert rel_err(z.detach().cpu().numpy(), z_ref.numpy()) < 1e-5
    z.sum().backward()
    assert alpha.grad is None and lin.weight.grad is not None


def test_adaptive_sage_use_neigh_gradients(cuda_device):
    """The intended model (use_neigh=True): gradients of alpha (K7) and h through the aggregation."""
    from dance_amd.cellgraph import NeighborSampler
    from dance_amd.nn.gnn import _SageAggregateFn
    n_cells, n_genes, d = 60, 30, 16
    x, g = _graph(n_cells, n_genes, d, 4, cuda_device)
    seeds = torch.arange(n_genes, n_genes + 20, device=cuda_device)
    _, _, blocks = NeighborSampler([-1]).sample(g, seeds)
    blk = blocks[0]
    h = blk.srcdata["features"].clone().requires_grad_(True)
    alpha = (torch.rand(n_genes + 2, 1, device=cuda_device) + 0.5).requires_grad_(True)
    dn = torch.randn(20, d, device=cuda_device)
    neigh = _SageAggregateFn.apply(h, alpha, blk)
    neigh.backward(dn)
    # float64 reference from the block's explicit edge list
    rp = blk.rowptr.cpu().numpy()
    e_dst = np.repeat(np.arange(20), np.diff(rp))
    e_src = blk.col.cpu().numpy()
    w = blk.val.cpu().numpy()
    cid = blk.srcdata["cell_id"].cpu().numpy()
    da_ref = osg.sage_alpha_grad(e_src, e_dst, w, cid, cid[:20], n_genes, h.detach().cpu().numpy(), dn.cpu().numpy())
    assert rel_err(alpha.grad.cpu().numpy().ravel(), da_ref) < 1e-4
    idx = osg.sage_alpha_index(cid[e_src], cid[:20][e_dst], n_genes)
    coef = alpha.detach().cpu().numpy().ravel()[idx] * w / np.maximum(np.diff(rp), 1)[e_dst]
    dh_ref = np.zeros((blk.number_of_src_nodes(), d))
    np.add.at(dh_ref, e_src, coef[:, None] * dn.cpu().numpy()[e_dst].astype(np.float64))
    assert rel_err(h.grad.cpu().numpy(), dh_ref) < 1e-5


def test_scdeepsort_fit_predict_small(cuda_device, tmp_path):
    """BASELINE config 1 (plumbing): synthetic cells with 4 expression programmes, PCACellFeatureGraph -> fit ->
    predict; logits equal the restated reference arithmetic on the CPU; accuracy is high; checkpoint round-trips."""
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    n_cells, n_genes, n_types = 1200, 300, 4
    types = rng.integers(0, n_types, n_cells)
    rates = rng.gamma(0.3, 1.0, (n_types, n_genes)) * 2
    x = rng.poisson(rates[types]).astype(np.float32)
    onehot = np.eye(n_types, dtype=np.float32)[types]
    data = Data(AnnDataLite(x, obsm={"cell_type": onehot}), train_size=1000)
    data.set_config(feature_channel=None, feature_channel_type="X")
    ScDeepSort.preprocessing_pipeline(n_components=32, log_level="WARNING")(data)
    assert data.config["label_channel"] == "cell_type"
    g = data.data.uns["CellFeatureGraph"]
    y = torch.from_numpy(types)
    train_nodes = torch.cat((torch.arange(n_genes), n_genes + torch.arange(1000)))
    test_nodes = torch.cat((torch.arange(n_genes), n_genes + torch.arange(1000, n_cells)))
    g_train, g_test = g.subgraph(train_nodes), g.subgraph(test_nodes)
    model = ScDeepSort(32, 16, 1, "synthetic", "blob", batch_size=256, device="cuda", save_root=tmp_path, verbose=False)
    model.fit(g_train, y[:1000], epochs=12, lr=1e-2, val_ratio=0.2)
    prob = model.predict_proba(g_test)
    assert prob.shape == (200, n_types) and np.allclose(prob.sum(1), 1, atol=1e-5)
    pred, unsure = model.predict(g_test, return_unsure=True)
    acc = model.score(g_test, onehot[1000:])
    assert acc == pytest.approx((pred == types[1000:]).mean()) and acc > 0.9
    # logits == Linear2(relu(Linear1(features[cells]))) — reference arithmetic (gnn.py:92-96, scdeepsort.py:84-88)
    sd = {k: v.cpu() for k, v in model.model.state_dict().items()}
    feats = g_test.ndata["features"].cpu()[n_genes:]
    hid = torch.relu(feats @ sd["layers.0.layers.1.weight"].t() + sd["layers.0.layers.1.bias"])
    ref_prob = torch.softmax(hid @ sd["linear.weight"].t() + sd["linear.bias"], -1).numpy()
    assert np.abs(prob - ref_prob).max() < 1e-5
    assert torch.all(sd["alpha"] == 1)  # never trained in the reference as written
    assert (tmp_path / "saved_models/single_modality/cell_type_annotation/pretrained/synthetic/models/synthetic-blob.pt").exists()


def test_full_graph_eval_equals_block_eval(cuda_device, tmp_path):
    """``full_graph_eval`` (one pass over the CSR rows of all cells) gives the logits / statistics of the reference's
    batch-by-batch loop over sampled blocks (scdeepsort.py:272-283,299-330), including the aggregated ``neigh``."""
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    n_cells, n_genes, d = 900, 70, 24
    x, g = _graph(n_cells, n_genes, d, 11, cuda_device)
    labels = torch.from_numpy(np.random.default_rng(0).integers(0, 5, n_cells))
    model = ScDeepSort(d, 16, 1, "synthetic", "fg", batch_size=128, device="cuda", save_root=tmp_path, verbose=False)
    model.shuffle_generator = torch.Generator().manual_seed(0)
    model.fit(g, labels, epochs=2, lr=1e-2)
    assert g.gene_prefix() == n_genes
    idx = torch.arange(n_genes + 100, n_genes + 700, device=cuda_device)
    gg = g.to("cuda")
    gg.ndata["label"] = torch.cat((-torch.ones(n_genes, dtype=torch.long), labels)).to(cuda_device)
    full = model.evaluate(gg, idx)
    neigh_full = model.model.layers[0].last_neigh.clone()
    prob_full = model.predict_proba(g)
    model.full_graph_eval = False
    blockwise = model.evaluate(gg, idx)
    prob_block = model.predict_proba(g)
    assert full == blockwise
    assert np.abs(prob_full - prob_block).max() < 1e-6
    # the full-graph neigh rows are those of the per-batch blocks (last batch of predict_proba = the last cells)
    last = model.model.layers[0].last_neigh
    assert torch.allclose(neigh_full[-last.shape[0]:], last, rtol=1e-5, atol=1e-6)


def test_two_layer_full_graph_eval(cuda_device, tmp_path):
    """Two AdaptiveSAGE layers (scdeepsort.py:183 ``[-1] * n_layers``): the full-graph pass updates every node in the inner layer —
    its gene rows through dh_sage_window_splitk, its cell rows through dh_sage_window_mfma — and gives the statistics / probabilities
    of the batch-by-batch loop over two-hop sampled blocks; the inner layer's aggregation equals the gather kernel on the whole CSR."""
    from dance_amd import kernels
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    n_cells, n_genes, d = 2600, 90, 24
    x, g = _graph(n_cells, n_genes, d, 5, cuda_device)
    labels = torch.from_numpy(np.random.default_rng(1).integers(0, 4, n_cells))
    model = ScDeepSort(d, 16, 2, "synthetic", "fg2", batch_size=512, device="cuda", save_root=tmp_path, verbose=False)
    model.shuffle_generator = torch.Generator().manual_seed(0)
    model.fit(g, labels, epochs=1, lr=1e-2)
    with torch.no_grad():
        model.model.alpha.copy_(torch.rand_like(model.model.alpha) + 0.5)  # (never trained in the reference: make the scales visible)
    gg = g.to("cuda")
    gg.ndata["label"] = torch.cat((-torch.ones(n_genes, dtype=torch.long), labels)).to(cuda_device)
    idx = torch.arange(n_genes + 50, n_genes + 2000, device=cuda_device)
    full = model.evaluate(gg, idx)
    prob_full = model.predict_proba(g)
    neigh0 = model.model.layers[0].last_neigh.clone()
    assert neigh0.shape == (n_genes + n_cells, d)
    cid = gg.ndata["cell_id"]
    ref = kernels.sage_aggregate(gg.rowptr, gg.col, gg.val, cid, cid, model.model.alpha.detach().reshape(-1).float(),
                                 gg.ndata["features"].contiguous())
    assert rel_err(neigh0[:n_genes].cpu().numpy(), ref[:n_genes].cpu().numpy()) < 2e-5     # gene rows: split-K
    assert rel_err(neigh0[n_genes:].cpu().numpy(), ref[n_genes:].cpu().numpy()) < 2e-5     # cell rows: gene window
    model.full_graph_eval = False
    assert model.evaluate(gg, idx) == full
    assert np.abs(prob_full - model.predict_proba(g)).max() < 1e-6


def test_scdeepsort_captured_step_equals_eager(cuda_device, tmp_path, monkeypatch):
    """ScDeepSort.fit with every full training batch replayed from ONE captured hipGraph (static-shape cell block, AdaptiveSAGE's
    discarded aggregation included, loss, backward, capturable Adam) ends with the parameters of the eager loop (same seeds: the
    split and the batch order come from ``shuffle_generator``), fp32 and bf16 storage; the aggregation of the static block equals
    the sampled block's."""
    from dance_amd.cellgraph import NeighborSampler, StaticCellBlock
    from dance_amd.modules.single_modality.cell_type_annotation import scdeepsort
    from dance_amd.nn import AdaptiveSAGE
    n_cells, n_genes, d = 700, 120, 32
    x, g = _graph(n_cells, n_genes, d, 3, cuda_device)
    labels = torch.from_numpy(np.random.default_rng(0).integers(0, 5, n_cells))
    # the static block's discarded aggregation == the sampled block's
    seeds = n_genes + torch.randperm(n_cells, generator=torch.Generator().manual_seed(0))[:64].to(cuda_device)
    _, _, blocks = NeighborSampler([-1]).sample(g, seeds, True)
    sb = StaticCellBlock(g, 64)
    sb.seeds.copy_(seeds)
    sb.rebuild()
    alpha = nn.Parameter(torch.rand(n_genes + 2, 1, device=cuda_device) + 0.5)
    layer = AdaptiveSAGE(d, 8, alpha, nn.Identity(), nn.ReLU(), nn.Identity()).to(cuda_device)
    layer(blocks[0], blocks[0].srcdata["features"])
    ref_neigh = layer.last_neigh.clone()
    layer(sb, sb.srcdata["features"])
    assert int(sb.bad) == 0 and rel_err(layer.last_neigh.cpu().numpy(), ref_neigh.cpu().numpy()) < 1e-5
    monkeypatch.setattr(scdeepsort, "HIPGRAPH_MIN_BATCHES", 1)
    monkeypatch.setattr(scdeepsort, "MINISTEP", False)  # the hipGraph path: the fallback for shapes the persistent step (tests/test_gpu_ministep.py) does not cover
    for cd, tol in (("fp32", 1e-5), ("bf16", 2e-2)):
        out = {}
        for on in (True, False, "split"):  # "split": two graphs per step with the (here: one-rank) gradient all-reduce between them
            monkeypatch.setattr(scdeepsort, "HIPGRAPH", bool(on))
            torch.manual_seed(7)
            m = scdeepsort.ScDeepSort(d, 16, 1, "synthetic", f"cap{on}{cd}", batch_size=64, device="cuda", save_root=tmp_path, verbose=False,
                                      compute_dtype=cd)
            m.capture_split = on == "split"
            m.shuffle_generator = torch.Generator().manual_seed(11)
            m.fit(g, labels, epochs=3, lr=1e-2, val_ratio=0.2)
            assert m._use_graph == bool(on) and (m._captured is not None) == bool(on)
            out[on] = ({k: v.detach().float().cpu().numpy() for k, v in m.model.state_dict().items()}, m.predict_proba(g))
        for k in out[True][0]:
            assert np.array_equal(out["split"][0][k], out[True][0][k]), (cd, k)
        for k in out[True][0]:
            assert rel_err(out[True][0][k], out[False][0][k]) < tol, (cd, k)
        assert np.abs(out[True][1] - out[False][1]).max() < tol * 10
