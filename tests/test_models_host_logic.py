"""Model-level host logic on CPU tensors against the reference's golden outputs: the ``fit`` / ``predict`` loops, loss
arithmetic, optimiser wiring and state-dict layout of the mirrored models, with every HIP kernel replaced by the torch /
scipy stand-ins of tests/cpu_ops.py.  The goldens come from the reference's OWN classes (tests/golden/make_golden.py); the
GPU suite runs the same comparisons through the kernels (tests/test_gpu_*.py).  Nothing here needs a GPU."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import cpu_ops
from conftest import rel_err

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture
def cpu_kernels(monkeypatch):
    from dance_amd import kernels
    for name in cpu_ops.STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    return kernels


def test_scdsc_fit_predict_vs_reference_on_cpu(cpu_kernels, tmp_path):
    """ScDSC.fit (pre-training, joint loop, best-ARI checkpoint; scdsc.py:200-288) == the reference's own fit."""
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    g = np.load(os.path.join(GOLDEN, "scdsc_fit.npz"))
    kw = json.loads(str(g["sf_kw"]))
    n = g["sf_x"].shape[0]
    m = ScDSC(pretrain_path=str(tmp_path / "ae.pt"), device="cpu", **kw)
    sd0 = {k.split("::", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sf_sd0::")}
    assert sorted(sd0) == sorted(m.model.state_dict())
    m.model.load_state_dict(sd0)
    adj = sp.csr_matrix((g["sf_adj_data"], g["sf_adj_indices"], g["sf_adj_indptr"]), shape=(n, n))
    torch.manual_seed(10)
    m.fit((adj, g["sf_x"], g["sf_counts"], g["sf_n_counts"].astype(np.float64)), g["sf_y"], lr=1e-3, epochs=12, pt_epochs=3, pt_batch_size=32,
          pt_lr=1e-3)
    q = m.predict_proba()
    assert q.shape == g["sf_q"].shape and np.allclose(q.sum(1), 1, atol=1e-5)
    assert rel_err(q, g["sf_q"]) < 5e-3
    assert (m.predict() == g["sf_pred"]).mean() > 0.98
    for k in g.files:
        if k.startswith("sf_sd1::") and "num_batches_tracked" not in k:
            got = m.model.state_dict()[k.split("::", 1)[1]].cpu().numpy()
            assert np.abs(got - g[k]).max() < 1.5e-2 * max(1.0, np.abs(g[k]).max()), k


SCTAG_KW = dict(n_clusters=3, k=3, hidden_dim=16, latent_dim=6, dec_dim=[12, 16, 20], dropout=0.0, device="cpu")


def test_sctag_forward_and_fit_vs_reference_on_cpu(cpu_kernels):
    """ScTAG forward (TAGConv hops, adjacency / ZINB decoders) and fit (pre-training + AMSGrad loop; sctag.py:84-528)."""
    from dance_amd.modules.single_modality.clustering.sctag import ScTAG
    g = np.load(os.path.join(GOLDEN, "sctag.npz"))
    m = ScTAG(**SCTAG_KW)
    m.init_model(g["tg_adj"], g["tg_x"])
    sd = {k.split("::", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("tg_sd0::")}
    assert sorted(sd) == sorted(m.state_dict())
    m.load_state_dict(sd)
    x = torch.from_numpy(g["tg_x"])
    with torch.no_grad():
        adj_out, z, q, mean, disp, pi = m.forward(m.g_n, x)
        enc_u = m.encoder1(m.g_n, x)
    for got, name in ((adj_out, "adj_out"), (z, "z"), (q, "q"), (mean, "mean"), (disp, "disp"), (pi, "pi"), (enc_u, "enc_unweighted")):
        assert rel_err(got.numpy(), g["tg_" + name]) < 1e-4, name
    torch.manual_seed(5)
    np.random.seed(0)
    m = ScTAG(**SCTAG_KW)
    m.fit((g["tg_adj"], g["tg_x"], g["tg_counts"], g["tg_n_counts"].astype(np.float64)), g["tg_y"], epochs=4, pretrain_epochs=3, lr=5e-3, w_d=0.1)
    q = m.predict_proba()
    assert rel_err(q, g["tg_fit_q"]) < 2e-2
    assert (m.predict() == g["tg_fit_pred"]).mean() > 0.95
    for k in g.files:
        if k.startswith("tg_sd1::"):
            got = m.state_dict()[k.split("::", 1)[1]].numpy()
            assert np.abs(got - g[k]).max() < 4e-2 * max(1.0, np.abs(g[k]).max()), k


def test_stagate_gatconv_and_pretrain_vs_reference_on_cpu(cpu_kernels):
    """GATConv forward / hand-written backward wiring and the tied-weight Stagate auto-encoder (stagate.py:31-330)."""
    from dance_amd.modules.spatial.spatial_domain.stagate import GATConv, Stagate
    g = np.load(os.path.join(GOLDEN, "stagate.npz"))
    d, c = g["sg_conv_lin"].shape
    conv = GATConv(d, c, heads=1, concat=False, dropout=0, add_self_loops=False, bias=False)
    with torch.no_grad():
        conv.lin_src.copy_(torch.from_numpy(g["sg_conv_lin"]))
        conv.att_src.copy_(torch.from_numpy(g["sg_conv_att_src"]))
        conv.att_dst.copy_(torch.from_numpy(g["sg_conv_att_dst"]))
    x = torch.from_numpy(g["sg_x"]).requires_grad_(True)
    ei = torch.from_numpy(g["sg_edge_index"])
    y, (ei2, alpha) = conv(x, ei, return_attention_weights=True)
    assert torch.equal(ei2, ei)
    assert rel_err(y.detach().numpy(), g["sg_conv_out"]) < 1e-5
    assert rel_err(alpha.detach().numpy(), g["sg_conv_alpha"]) < 1e-5
    y.backward(torch.from_numpy(g["sg_conv_dy"]))
    assert rel_err(x.grad.numpy(), g["sg_conv_dx"]) < 1e-4
    assert rel_err(conv.lin_src.grad.numpy(), g["sg_conv_dlin"]) < 1e-4
    assert rel_err(conv.att_src.grad.numpy(), g["sg_conv_datt_src"]) < 1e-4
    assert rel_err(conv.att_dst.grad.numpy(), g["sg_conv_datt_dst"]) < 1e-4
    m = Stagate([g["sg_x"].shape[1], 12, 6], device="cpu")
    sd = {k.split("::", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sg_sd0::")}
    assert sorted(sd) == sorted(m.state_dict())
    m.load_state_dict(sd)
    with torch.no_grad():
        h2, h4 = m(torch.from_numpy(g["sg_x"]), ei)
    assert rel_err(h2.numpy(), g["sg_h2"]) < 1e-5 and rel_err(h4.numpy(), g["sg_h4"]) < 1e-5
    m.pretrain(g["sg_x"], g["sg_edge_index"], lr=1e-2, weight_decay=1e-4, epochs=5, gradient_clipping=5)
    assert rel_err(m.rep, g["sg_rep"]) < 5e-3


def _scheteronet(gold):
    import types
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import scHeteroNet
    n, d, c, hid = (int(v) for v in gold["sh_dims"])
    m = scHeteroNet(d, c, torch.from_numpy(gold["sh_edge_index"]), n, hid, 2, 0.0, True, "cpu", 100.0)
    sd = {k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sh_sd0::")}
    assert sorted(sd) == sorted(m.state_dict())
    m.load_state_dict(sd)
    ds = types.SimpleNamespace(x=torch.from_numpy(gold["sh_x"]), edge_index=torch.from_numpy(gold["sh_edge_index"]),
                               y=torch.from_numpy(gold["sh_y"])[:, None], splits={"train": torch.arange(0, n, 2)}, node_idx=torch.arange(n))
    return m, ds


def test_scheteronet_forward_and_fit_step_vs_reference_on_cpu(cpu_kernels):
    """init_adj (gcn_norm, strict two-hop pattern), HetConv forward, energy propagation, detect and one fit step
    (scheteronet.py:374-789)."""
    import types
    gold = np.load(os.path.join(GOLDEN, "scheteronet.npz"))
    m, ds = _scheteronet(gold)
    enc = m.encoder
    assert rel_err(enc.adj_t.to_dense().numpy(), gold["sh_adj_t"]) < 1e-6
    assert np.array_equal(enc.adj_t2.to_dense().numpy() != 0, gold["sh_adj_t2"] != 0)
    assert rel_err(enc.adj_t2.to_dense().numpy(), gold["sh_adj_t2"]) < 1e-6
    m.eval()
    with torch.no_grad():
        h, mean, disp, pi = enc(ds.x, ds.edge_index, decoder=True)
        assert rel_err(h.numpy(), gold["sh_logits"]) < 1e-4
        assert rel_err(mean.numpy(), gold["sh_mean"]) < 1e-4 and rel_err(disp.numpy(), gold["sh_disp"]) < 1e-4
        assert rel_err(pi.numpy(), gold["sh_pi"]) < 1e-4
        e = torch.from_numpy(gold["sh_e"])
        assert rel_err(m.propagation(e, ds.edge_index, 2, 0.5).numpy(), gold["sh_prop"]) < 1e-5
        assert rel_err(m.two_hop_propagation(e, ds.edge_index, 1, 0.3).numpy(), gold["sh_prop2"]) < 1e-5
        assert rel_err(m.detect(ds, ds.node_idx, "cpu", 1.0, True, False, 2, 0.5).numpy(), gold["sh_detect"]) < 1e-4
    assert np.array_equal(m.predict(ds).numpy(), gold["sh_logits"].argmax(1))
    m, ds = _scheteronet(gold)
    adata = types.SimpleNamespace(raw=types.SimpleNamespace(X=gold["sh_counts"]), obs={"size_factors": gold["sh_size_factors"]})
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    loss = m.fit(ds, ds, True, adata, 0.5, 0.0, 0.4, torch.nn.NLLLoss(), opt)
    assert abs(float(loss) - float(gold["sh_loss"])) < 1e-4 * abs(float(gold["sh_loss"]))
    for k in gold.files:
        if k.startswith("sh_sd1::") and "num_batches_tracked" not in k:
            got = m.state_dict()[k.split("::", 1)[1]].numpy()
            assert np.abs(got - gold[k]).max() < 2e-3 * max(1.0, np.abs(gold[k]).max()), k


def test_spagcn_heads_vs_reference_on_cpu(cpu_kernels):
    """SimpleGCDEC forward / target distribution / KL loss / gradients vs the reference's class (spagcn.py:369-478), the
    calculate_p / search_l bisection and calc_adj_exp vs the oracle restatement (spagcn.py:249-334), and a planted-domain fit
    (kmeans init, dense adjacency) through the host loop."""
    from oracle import matrix as om
    from oracle import spagcn as osp
    from sklearn.metrics import adjusted_rand_score
    from dance_amd.modules.spatial.spatial_domain.spagcn import SimpleGCDEC, SpaGCN, calculate_p
    h = np.load(os.path.join(GOLDEN, "model_heads.npz"))
    m = SimpleGCDEC(12, 12, device="cpu")
    t = torch.from_numpy
    m.gc.weight.data, m.gc.bias.data = t(h["gcdec_w"]), t(h["gcdec_b"])
    m.mu = torch.nn.Parameter(t(h["gcdec_mu"]))
    z, q = m.forward(t(h["gcdec_x"]), t(h["gcdec_adj"]))
    p = m.target_distribution(q)
    loss = m.loss_function(p.data, q)
    loss.backward()
    assert rel_err(z.detach().numpy(), h["gcdec_z"]) < 1e-5 and rel_err(q.detach().numpy(), h["gcdec_q"]) < 1e-5
    assert rel_err(p.detach().numpy(), h["gcdec_p"]) < 1e-5
    assert abs(loss.item() - float(h["gcdec_loss"])) < 1e-5 * max(1, abs(float(h["gcdec_loss"])))
    assert rel_err(m.gc.weight.grad.numpy(), h["gcdec_dw"]) < 1e-3 and rel_err(m.mu.grad.numpy(), h["gcdec_dmu"]) < 1e-3
    rng = np.random.default_rng(0)
    xyz = (rng.random((300, 3)) * 40).astype(np.float32)
    adj = om.pairwise_distance(xyz, 0)
    for l in (0.5, 2.0, 10.0):
        assert abs(calculate_p(adj, l, "cpu") - osp.calculate_p(adj, l)) < 1e-4 * max(1.0, osp.calculate_p(adj, l))
    model = SpaGCN(device="cpu")
    l_hip, l_ref = model.search_l(0.5, adj, start=0.01, end=1000, tol=0.01, max_run=100), osp.search_l(0.5, adj)
    assert l_ref is not None and l_hip == pytest.approx(l_ref, rel=1e-6)
    model.set_l(l_hip)
    assert rel_err(model.calc_adj_exp(adj).numpy(), osp.calc_adj_exp(adj, l_hip)) < 1e-5
    torch.manual_seed(0)
    rng = np.random.default_rng(1)
    side = 16
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    dom = (xy[:, 0] >= side / 2).astype(int) + 2 * (xy[:, 1] >= side / 2).astype(int)
    embed = (np.eye(4)[dom] @ rng.standard_normal((4, 10)) * 2 + rng.standard_normal((xy.shape[0], 10))).astype(np.float32)
    adj = om.pairwise_distance(xy, 0)
    model = SpaGCN(device="cpu")
    model.set_l(model.search_l(0.5, adj))
    np.random.seed(0)
    model.fit((embed, adj), init="kmeans", n_clusters=4, epochs=40, lr=0.01, tol=1e-4)
    assert adjusted_rand_score(dom, model.predict((embed, adj))) > 0.9
    with pytest.raises(ValueError):
        SpaGCN(device="cpu").fit((embed, adj))


def test_scdsc_model_forward_vs_reference_on_cpu(cpu_kernels):
    """ScDSCModel.forward (AE + four GNNLayers + ZINB heads; scdsc.py:291-472) with the reference's weights."""
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSCModel
    h = np.load(os.path.join(GOLDEN, "model_heads.npz"))
    kw = json.loads(str(h["scdsc_kw"]))
    model = ScDSCModel(**kw, device="cpu").eval()
    sd = {k.split("::", 1)[1]: torch.from_numpy(h[k]) for k in h.files if k.startswith("scdsc_sd::")}
    assert set(sd) == set(model.state_dict())
    model.load_state_dict(sd)
    n = h["scdsc_x"].shape[0]
    a = sp.csr_matrix((h["scdsc_adj_data"], h["scdsc_adj_indices"], h["scdsc_adj_indptr"]), shape=(n, n)).tocoo()
    adj = torch.sparse_coo_tensor(np.vstack((a.row, a.col)).astype(np.int64), a.data, (n, n))
    with torch.no_grad():
        x_bar, q, predict, z3, _mean, _disp, _pi, zinb = model(torch.from_numpy(h["scdsc_x"]), adj)
    for name, val in (("x_bar", x_bar), ("q", q), ("predict", predict), ("z3", z3), ("mean", _mean), ("disp", _disp), ("pi", _pi)):
        assert rel_err(val.numpy(), h[f"scdsc_{name}"]) < 1e-5, name


def test_scdsc_raw_heads_is_the_same_loss_function_on_cpu(cpu_kernels):
    """``ScDSCModel._forward(x, adj, raw_heads=True)`` (what the joint loop trains through): the heads' Linear outputs + ``ZINBLoss.from_logits`` give
    the value and the parameter gradients of the reference's composition (heads with MeanAct / DispAct / Sigmoid, then ZINBLoss,
    scdsc.py:409-411, :463-465, :279-283); everything else in the returned tuple is unchanged."""
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSCModel
    h = np.load(os.path.join(GOLDEN, "model_heads.npz"))
    kw = json.loads(str(h["scdsc_kw"]))
    model = ScDSCModel(**kw, device="cpu").train()
    model.load_state_dict({k.split("::", 1)[1]: torch.from_numpy(h[k]) for k in h.files if k.startswith("scdsc_sd::")})
    n = h["scdsc_x"].shape[0]
    a = sp.csr_matrix((h["scdsc_adj_data"], h["scdsc_adj_indices"], h["scdsc_adj_indptr"]), shape=(n, n)).tocoo()
    adj = torch.sparse_coo_tensor(np.vstack((a.row, a.col)).astype(np.int64), a.data, (n, n))
    x = torch.from_numpy(h["scdsc_x"])
    counts = torch.poisson(torch.rand(n, kw["n_input"], generator=torch.Generator().manual_seed(0)) * 2)
    sf = torch.rand(n, dtype=torch.float64, generator=torch.Generator().manual_seed(1)) + 0.5
    heads = [p for mod in (model._dec_mean, model._dec_disp, model._dec_pi) for p in mod.parameters()]
    out = {}
    for raw in (False, True):
        model.ae._cache = None
        res = model._forward(x, adj, raw)
        loss = res[-1](counts, res[4], res[5], res[6], sf)
        out[raw] = (res, float(loss), torch.autograd.grad(loss, heads))
    (r0, l0, g0), (r1, l1, g1) = out[False], out[True]
    assert abs(l0 - l1) < 1e-6 * abs(l0)
    for a_, b_ in zip(g0, g1):
        assert rel_err(b_.numpy(), a_.numpy()) < 1e-5
    for i in range(4):
        assert torch.equal(r0[i], r1[i])
    act = (torch.clamp(torch.exp(r1[4]), 1e-5, 1e6), torch.clamp(torch.nn.functional.softplus(r1[5]), 1e-4, 1e4), torch.sigmoid(r1[6]))
    for i in range(3):
        assert torch.allclose(act[i], r0[4 + i], rtol=1e-6, atol=0)


def test_scdeepsort_fit_predict_small_on_cpu(cpu_kernels, tmp_path):
    """BASELINE config 1 through the host logic: PCACellFeatureGraph -> ScDeepSort.fit (train / validation split, block
    loader, AdaptiveSAGE with the computed-and-dropped ``neigh``, checkpointing) -> predict with the unsure rule; the logits
    equal the reference arithmetic (gnn.py:92-96, scdeepsort.py:84-88) evaluated in plain torch."""
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    n_cells, n_genes, n_types, n_train = 360, 80, 4, 300
    types = rng.integers(0, n_types, n_cells)
    rates = rng.gamma(0.3, 1.0, (n_types, n_genes)) * 2
    x = rng.poisson(rates[types]).astype(np.float32)
    onehot = np.eye(n_types, dtype=np.float32)[types]
    data = Data(AnnDataLite(x, obsm={"cell_type": onehot}), train_size=n_train)
    data.set_config(feature_channel=None, feature_channel_type="X")
    pipe = ScDeepSort.preprocessing_pipeline(n_components=16, log_level="WARNING")
    for t in getattr(pipe, "transforms", []):
        if hasattr(t, "device"):
            t.device = "cpu"
    pipe(data)
    g = data.data.uns["CellFeatureGraph"]
    train_nodes = torch.cat((torch.arange(n_genes), n_genes + torch.arange(n_train)))
    test_nodes = torch.cat((torch.arange(n_genes), n_genes + torch.arange(n_train, n_cells)))
    g_train, g_test = g.subgraph(train_nodes), g.subgraph(test_nodes)
    model = ScDeepSort(16, 16, 1, "synthetic", "blob", batch_size=64, device="cpu", save_root=tmp_path, verbose=False)
    model.fit(g_train, torch.from_numpy(types)[:n_train], epochs=15, lr=1e-2, val_ratio=0.2)
    prob = model.predict_proba(g_test)
    assert prob.shape == (n_cells - n_train, n_types) and np.allclose(prob.sum(1), 1, atol=1e-5)
    pred, unsure = model.predict(g_test, return_unsure=True)
    acc = model.score(g_test, onehot[n_train:])
    assert acc == pytest.approx((pred == types[n_train:]).mean()) and acc > 0.85
    sd = model.model.state_dict()
    feats = g_test.ndata["features"][n_genes:]
    hid = torch.relu(feats @ sd["layers.0.layers.1.weight"].t() + sd["layers.0.layers.1.bias"])
    ref_prob = torch.softmax(hid @ sd["linear.weight"].t() + sd["linear.bias"], -1).numpy()
    assert np.abs(prob - ref_prob).max() < 1e-5
    assert torch.all(sd["alpha"] == 1)
    assert model.model.layers[0].last_neigh is not None   # the aggregation ran (and was dropped, as in the reference)
    assert (tmp_path / "saved_models/single_modality/cell_type_annotation/pretrained/synthetic/models/synthetic-blob.pt").exists()


def test_graph_transforms_vs_reference_on_cpu(cpu_kernels):
    """The graph transforms' host logic (CSR views of X, edge-id bookkeeping that restores the reference's edge order, node
    frames, kNN / radius edge lists, the histology z coordinate) against tests/golden/graph_builders.npz — outputs of the
    reference's own ``__call__`` / ``build_graph`` methods."""
    from oracle import matrix as om
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import CellFeatureGraph, HeteronetGraph, SpaGCNGraph, StagateGraph
    gold = np.load(os.path.join(GOLDEN, "graph_builders.npz"))
    for norm in (0, 1):
        data = Data(AnnDataLite(gold["cfg_x"], obsm={"f": gold["cfg_cell_feat"]}, varm={"f": gold["cfg_gene_feat"]}))
        t = CellFeatureGraph("f", normalize_edges=bool(norm))
        t.device = "cpu"
        t(data)
        g = data.data.uns["CellFeatureGraph"]
        tag = f"cfg_norm{norm}_"
        src, dst = g.edges()
        assert np.array_equal(src.numpy(), gold[tag + "src"]) and np.array_equal(dst.numpy(), gold[tag + "dst"])
        assert rel_err(g.edata["weight"].numpy().ravel(), gold[tag + "weight"]) < 1e-6
        assert np.array_equal(g.ndata["cell_id"].numpy(), gold[tag + "cell_id"])
        assert np.array_equal(g.ndata["feat_id"].numpy(), gold[tag + "feat_id"])
        assert np.array_equal(g.ndata["features"].numpy(), gold[tag + "features"])
    het = HeteronetGraph(knn_num=5)
    het.device = "cpu"
    assert np.array_equal(het.build_graph(gold["het_feats"], knears=5), gold["het_edges"])
    xy = gold["stg_xy"]
    data = Data(AnnDataLite(np.zeros((xy.shape[0], 2), np.float32), obsm={"spatial_pixel": xy}))
    for t in (StagateGraph("radius", radius=1.7, out="r"), StagateGraph("knn", n_neighbors=4, out="k")):
        t.device = "cpu"
        t(data)
    assert np.array_equal(np.asarray(data.data.obsp["r"].todense(), dtype=np.float32), gold["stg_radius"])
    assert np.array_equal(np.asarray(data.data.obsp["k"].todense(), dtype=np.float32), gold["stg_knn"])
    n = gold["spg_xy"].shape[0]
    data = Data(AnnDataLite(np.zeros((n, 2), np.float32), obsm={"spatial": gold["spg_xy"], "spatial_pixel": gold["spg_xy_pixel"]},
                            uns={"image": gold["spg_img"]}))
    t = SpaGCNGraph(alpha=float(gold["spg_alpha"]), beta=int(gold["spg_beta"]))
    t.device = "cpu"
    t(data)
    assert np.array_equal(data.data.obsp["SpaGCNGraph"], om.pairwise_distance(gold["spg_xyz"], 0))


def test_adaptive_sage_gradient_wiring_on_cpu(cpu_kernels):
    """The intended AdaptiveSAGE model (use_neigh=True): the hand-written backward of the aggregation — alpha gradient through
    the SDDMM + float64 bins, h gradient through the transposed block — against the float64 restatement (oracle/sage.py)."""
    from oracle import sage as osg
    from dance_amd.cellgraph import CellGeneGraph, NeighborSampler
    from dance_amd.nn.gnn import _SageAggregateFn
    import test_graphsc_host_logic as gh
    gold = np.load(gh.GOLD)
    g = gh._graph(gold)
    n_cells, n_genes = gold["gsc_x"].shape
    d = g.ndata["features"].shape[1]
    seeds = torch.arange(n_genes, n_genes + 12)
    _, _, blocks = NeighborSampler([-1]).sample(g, seeds)
    blk = blocks[0]
    torch.manual_seed(0)
    feats = torch.randn(blk.number_of_src_nodes(), 8)
    h = feats.clone().requires_grad_(True)
    alpha = (torch.rand(n_genes + 2, 1) + 0.5).requires_grad_(True)
    dn = torch.randn(12, 8)
    neigh = _SageAggregateFn.apply(h, alpha, blk)
    neigh.backward(dn)
    rp = blk.rowptr.numpy()
    e_dst = np.repeat(np.arange(12), np.diff(rp))
    e_src, w = blk.col.numpy(), blk.val.numpy()
    cid = blk.srcdata["cell_id"].numpy()
    assert rel_err(neigh.detach().numpy(), osg.sage_neigh(e_src, e_dst, w, cid, cid[:12], alpha.detach().numpy(), feats.numpy(), 12)) < 1e-5
    da_ref = osg.sage_alpha_grad(e_src, e_dst, w, cid, cid[:12], n_genes, feats.numpy(), dn.numpy())
    assert rel_err(alpha.grad.numpy().ravel(), da_ref) < 1e-4
    idx = osg.sage_alpha_index(cid[e_src], cid[:12][e_dst], n_genes)
    coef = alpha.detach().numpy().ravel()[idx] * w / np.maximum(np.diff(rp), 1)[e_dst]
    dh_ref = np.zeros((blk.number_of_src_nodes(), 8))
    np.add.at(dh_ref, e_src, coef[:, None] * dn.numpy()[e_dst].astype(np.float64))
    assert rel_err(h.grad.numpy(), dh_ref) < 1e-5


def test_sctag_scalable_adjacency_decoder_on_cpu(cpu_kernels):
    """ScTAG(adj_dim=d): the adjacency decoder with an N-independent width and its loss over all N^2 pairs without an N x N matrix
    (autograd.adj_reconstruction_mse: dh_gram_pairwise_f32 + dh_sddmm_csr_f32) == the dense formula of sctag.py:470-471, :254 on the
    same factor, value and gradients; pretrain + fit run end to end on a SPARSE adjacency."""
    import torch.nn.functional as F
    from dance_amd.modules.single_modality.clustering.sctag import ScTAG
    g = np.load(os.path.join(GOLDEN, "sctag.npz"))
    adj, x = g["tg_adj"], g["tg_x"]
    torch.manual_seed(1)
    m = ScTAG(n_clusters=3, k=3, hidden_dim=16, latent_dim=6, dec_dim=[12, 16, 20], dropout=0.0, device="cpu", adj_dim=8)
    m.init_model(sp.csr_matrix(adj), x)
    assert m.decoder_adj.dec_1.weight.shape == (8, 6) and m.adj_target.nnz == int((adj != 0).sum())
    xt = torch.from_numpy(x)
    z0, z, q, mean, disp, pi = m.forward(m.g_n, xt)
    assert z0.shape == (adj.shape[0], 8)
    loss = m.adj_loss(z0, None)
    gw, = torch.autograd.grad(loss, m.decoder_adj.dec_1.weight, retain_graph=True)
    zd = z0.detach().double().requires_grad_(True)
    ref = torch.mean(F.mse_loss(torch.sigmoid(zd @ zd.t()), torch.from_numpy(adj).double()))
    assert abs(float(loss) - float(ref)) < 1e-6 * abs(float(ref))
    gz, = torch.autograd.grad(ref, zd)
    gz0, = torch.autograd.grad(loss, z0)
    assert rel_err(gz0.numpy(), gz.numpy()) < 1e-5 and float(gw.abs().sum()) > 0
    # the dense mode on the same adjacency builds the same graphs (TAGConv inputs) as the sparse one
    m2 = ScTAG(n_clusters=3, k=3, hidden_dim=16, latent_dim=6, dec_dim=[12, 16, 20], dropout=0.0, device="cpu")
    m2.init_model(adj, x)
    assert torch.equal(m2.g_n.csr.col, m.g_n.csr.col) and rel_err(m2.g_n.csr.val.numpy(), m.g_n.csr.val.numpy()) < 1e-6
    torch.manual_seed(5)
    np.random.seed(0)
    m = ScTAG(n_clusters=3, k=3, hidden_dim=16, latent_dim=6, dec_dim=[12, 16, 20], dropout=0.1, device="cpu", adj_dim=8)
    m.fit((sp.csr_matrix(adj), x, g["tg_counts"], g["tg_n_counts"].astype(np.float64)), g["tg_y"], epochs=3, pretrain_epochs=3, lr=5e-3)
    assert m.predict().shape == (adj.shape[0], ) and np.isfinite(m.predict_proba()).all()


def check_gc_dec(device):
    """GC_DEC (spagcn.py:588-697, the two-layer DEC model) against the reference's own class run on torch-CPU (tests/golden/gc_dec.npz):
    forward, target distribution, KL loss, every gradient, and ``fit_with_init`` (5 SGD epochs, centres from given labels)."""
    from dance_amd.modules.spatial.spatial_domain.spagcn import GC_DEC
    g = np.load(os.path.join(GOLDEN, "gc_dec.npz"))
    fin, h1, h2, k = (int(v) for v in g["gd_dims"])
    m = GC_DEC(fin, h1, h2, n_clusters=k, dropout=0.0, alpha=0.2, device=device)
    m.load_state_dict({key.split("::", 1)[1]: torch.from_numpy(g[key]) for key in g.files if key.startswith("gd_sd::")})
    m.to(device)
    x, adj = torch.from_numpy(g["gd_x"]).to(device), torch.from_numpy(g["gd_adj"]).to(device)
    z, q = m(x, adj)
    p = m.target_distribution(q).data
    loss = m.loss_function(p, q)
    loss.backward()
    assert rel_err(z.detach().cpu().numpy(), g["gd_z"]) < 1e-5 and rel_err(q.detach().cpu().numpy(), g["gd_q"]) < 1e-5
    assert rel_err(p.cpu().numpy(), g["gd_p"]) < 1e-5 and abs(loss.item() - float(g["gd_loss"])) < 1e-5 * max(1.0, abs(float(g["gd_loss"])))
    for name, par in m.named_parameters():
        assert rel_err(par.grad.cpu().numpy(), g[f"gd_grad::{name}"]) < 1e-3, name
    m.zero_grad()
    m.fit_with_init(g["gd_x"], g["gd_adj"], g["gd_init_y"], lr=0.01, epochs=5, update_interval=2, opt="sgd")
    zf, qf = m.predict(g["gd_x"], g["gd_adj"])
    assert rel_err(zf.detach().cpu().numpy(), g["gd_fit_z"]) < 1e-4 and rel_err(qf.detach().cpu().numpy(), g["gd_fit_q"]) < 1e-4
    for key in g.files:
        if key.startswith("gd_fit_sd::"):
            assert rel_err(m.state_dict()[key.split("::", 1)[1]].cpu().numpy(), g[key]) < 1e-4, key


def test_gc_dec_vs_reference_on_cpu(cpu_kernels):
    check_gc_dec("cpu")


@pytest.mark.parametrize("tag,block_eval", [("one", True), ("mb", True), ("peak", True), ("one", False)])
def test_scdeepsort_fit_loop_vs_reference_on_cpu(cpu_kernels, tmp_path, monkeypatch, tag, block_eval):
    """ScDeepSort.fit / cal_loss / evaluate / predict_proba / predict == the reference's own loop (scdeepsort.py:142-349 run over
    the DGL stub, tests/golden/scdeepsort.npz): split, per-epoch loss and (correct, unsure, accuracy), best-validation
    checkpoint, probabilities, unsure flags.  ``block_eval=False`` = the product's default single-pass evaluation; with one
    batch per epoch ("one") the seed order only permutes a sum, so it must land on the same numbers."""
    import scdeepsort_golden_checks as chk
    gold, kw = chk.load()
    chk.check_case(gold, kw, tag, "cpu", tmp_path, monkeypatch, block_eval=block_eval, rel_err=rel_err)


def test_scdsc_frozen_autoencoder_cache(cpu_kernels):
    """AE.forward hands back the first call's outputs while the autoencoder is frozen, in training mode and fed the same tensor — and
    moves the BatchNorm running statistics exactly as the recomputation would (the eval-mode passes of scdsc.py:256-263 read them);
    anything that could change the outputs (a parameter update, another input, eval mode, a trainable autoencoder) recomputes."""
    from dance_amd.modules.single_modality.clustering.scdsc import AE
    torch.manual_seed(0)
    kw = dict(n_enc_1=24, n_enc_2=16, n_enc_3=16, n_dec_1=16, n_dec_2=16, n_dec_3=24, n_input=20, n_z1=16, n_z2=12, n_z3=8)
    a, b = AE(**kw), AE(**kw)
    b.load_state_dict(a.state_dict())
    b.cache_frozen = False
    for m in (a, b):
        for p in m.parameters():
            p.requires_grad_(False)
        m.train()
    x = torch.randn(50, 20)
    for _ in range(4):
        oa, ob = a(x), b(x)
    assert a._cache is not None and b._cache is None
    assert all(torch.equal(u, v) for u, v in zip(oa, ob))                       # the cached tensors ARE the recomputed ones
    assert oa[0] is a(x)[0]
    b(x)
    for i in range(1, 10):
        ba, bb = getattr(a, f"BN{i}"), getattr(b, f"BN{i}")
        assert int(ba.num_batches_tracked) == int(bb.num_batches_tracked) == 5
        assert rel_err(ba.running_mean.numpy(), bb.running_mean.numpy()) < 1e-6 and rel_err(ba.running_var.numpy(), bb.running_var.numpy()) < 1e-6
    a.eval(), b.eval()
    kept = a._cache
    assert rel_err(a(x)[0].numpy(), b(x)[0].numpy()) < 1e-5                     # eval mode: running statistics, never the cache
    assert a._cache is kept                                                      # ... and it does not drop the kept outputs
    a.train(), b.train()
    assert a(x)[0] is oa[0]                                                      # the training pass after an evaluation pass: still the kept tensors
    b(x)
    with torch.no_grad():
        a.enc_1.weight.mul_(1.01)
        b.enc_1.weight.mul_(1.01)
    assert all(torch.equal(u, v) for u, v in zip(a(x), b(x)))                   # a parameter changed in place: recomputed
    y = torch.randn(50, 20)
    assert all(torch.equal(u, v) for u, v in zip(a(y), b(y)))                   # another input: recomputed
    a.enc_1.weight.requires_grad_(True)
    a(y)
    assert a._cache is None                                                      # a trainable autoencoder is never cached


def test_cross_entropy_sum_module_is_torchs_loss(cpu_kernels):
    """``autograd.CrossEntropySum`` (what ScDeepSort.fit uses for scdeepsort.py:185) == ``nn.CrossEntropyLoss(reduction="sum")``: value,
    gradient through an upstream factor, ``ignore_index`` rows, a non-fp32 input; soft targets fall through to torch."""
    from dance_amd.autograd import CrossEntropySum
    g = torch.Generator().manual_seed(5)
    x = torch.randn(97, 7, generator=g, requires_grad=True)
    y = torch.randint(0, 7, (97, ), generator=g)
    y[::9] = -100
    ours = CrossEntropySum()(x, y)
    gx, = torch.autograd.grad(ours * 0.5, x)
    xr = x.detach().clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss(reduction="sum")(xr, y)
    gr, = torch.autograd.grad(ref * 0.5, xr)
    assert torch.allclose(ours, ref, rtol=1e-6) and torch.allclose(gx, gr, rtol=1e-5, atol=1e-7)
    assert bool((gx[::9] == 0).all())
    xh = x.detach().to(torch.bfloat16).requires_grad_(True)
    lh = CrossEntropySum()(xh, y)
    gh, = torch.autograd.grad(lh, xh)
    assert gh.dtype == torch.bfloat16 and torch.allclose(lh, torch.nn.functional.cross_entropy(xh.float(), y, reduction="sum"), rtol=1e-5)
    soft = torch.softmax(torch.randn(97, 7, generator=g), 1)
    assert torch.allclose(CrossEntropySum()(x, soft), torch.nn.functional.cross_entropy(x, soft, reduction="sum"))
