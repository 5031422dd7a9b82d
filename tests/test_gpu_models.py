"""GPU: graph-sc, SpaGCN and scDSC model mirrors vs the oracle / reference-generated golden vectors."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from conftest import GOLDEN, rel_err
from oracle import graphs as og
from oracle import matrix as om
from oracle import sage as osg
from oracle import spagcn as osp

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def heads():
    return dict(np.load(os.path.join(GOLDEN, "model_heads.npz")))


def _cellgene_graph(n_cells, n_genes, d, seed, normalize_edges=False):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import CellFeatureGraph
    rng = np.random.default_rng(seed)
    x = ((rng.random((n_cells, n_genes)) < 0.25) * rng.uniform(0.1, 1, (n_cells, n_genes))).astype(np.float32)
    data = Data(AnnDataLite(x, obsm={"f": rng.standard_normal((n_cells, d)).astype(np.float32)},
                            varm={"f": rng.standard_normal((n_genes, d)).astype(np.float32)}))
    CellFeatureGraph("f", normalize_edges=normalize_edges)(data)
    return x, data.data.uns["CellFeatureGraph"]


@pytest.mark.parametrize("agg", ["sum", "mean"])
def test_weighted_graph_conv_block(cuda_device, agg):
    from dance_amd.cellgraph import MultiLayerFullNeighborSampler
    from dance_amd.modules.single_modality.clustering.graphsc import WeightedGraphConv
    x, g = _cellgene_graph(80, 40, 20, 1)
    seeds = torch.arange(40 + 10, 40 + 42, device=cuda_device)
    _, _, blocks = MultiLayerFullNeighborSampler(1).sample(g, seeds)
    blk = blocks[0]
    conv = WeightedGraphConv(20, 12, activation=F.relu).to(cuda_device)
    with torch.no_grad():
        conv.bias.uniform_(-0.2, 0.2)
    feat = blk.srcdata["features"].clone().requires_grad_(True)
    out = conv(blk, feat, agg=agg)
    rp = blk.rowptr.cpu().numpy()
    e_dst, e_src, w = np.repeat(np.arange(32), np.diff(rp)), blk.col.cpu().numpy(), blk.val.cpu().numpy()
    ref = osg.weighted_graph_conv(e_src, e_dst, w, feat.detach().cpu().numpy(), 32, conv.weight.detach().cpu().numpy(),
                                  conv.bias.detach().cpu().numpy(), agg=agg, activation="relu")
    assert rel_err(out.detach().cpu().numpy(), ref) < TOL
    # gradients vs a float64 dense restatement under torch autograd
    dy = torch.randn_like(out)
    out.backward(dy)
    S, D = blk.number_of_src_nodes(), 32
    a = torch.zeros(D, S, dtype=torch.float64)
    a.index_put_((torch.from_numpy(e_dst), torch.from_numpy(e_src)), torch.from_numpy(w).double(), accumulate=True)
    out_deg = torch.bincount(torch.from_numpy(e_src), minlength=S).clamp(min=1).double()
    in_deg = torch.from_numpy(np.diff(rp)).clamp(min=1).double()
    f64 = feat.detach().cpu().double().requires_grad_(True)
    w64 = conv.weight.detach().cpu().double().requires_grad_(True)
    b64 = conv.bias.detach().cpu().double().requires_grad_(True)
    r = a @ ((f64 * out_deg.pow(-0.5)[:, None]) @ w64)
    if agg == "mean":
        r = r / in_deg[:, None]
    r = torch.relu(r * in_deg.pow(-0.5)[:, None] + b64)
    r.backward(dy.cpu().double())
    assert rel_err(conv.weight.grad.cpu().numpy(), w64.grad.numpy()) < TOL
    assert rel_err(conv.bias.grad.cpu().numpy(), b64.grad.numpy()) < TOL
    assert rel_err(feat.grad.cpu().numpy(), f64.grad.numpy()) < TOL


def test_graphsc_fit_predict(cuda_device):
    from dance_amd.cellgraph import MultiLayerFullNeighborSampler
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC, block_dst_adjacency
    torch.manual_seed(0)
    x, g = _cellgene_graph(300, 60, 16, 2)
    _, _, blocks = MultiLayerFullNeighborSampler(1).sample(g, torch.arange(60, 60 + 50, device=cuda_device))
    adj = block_dst_adjacency(blocks[-1])
    assert torch.equal(adj, torch.eye(50, device=cuda_device))  # cells only link to genes + their own self loop
    model = GraphSC(in_feats=16, hidden_dim=24, hidden_1=12, n_clusters=3, device="cuda")
    assert sorted(model.model.state_dict()) == ["encoder.0.bias", "encoder.0.weight", "layer1.bias", "layer1.weight"]
    model.fit(g, epochs=3, lr=1e-3, batch_size=64)
    assert model.get_latent().shape == (300, 12) and np.isfinite(model.losses).all()
    assert model.losses[-1] < model.losses[0]
    pred = model.predict()
    assert pred.shape == (300, ) and set(pred) <= {0, 1, 2}
    model.cluster_method = "leiden"  # graphsc.py:264-265 run_leiden
    pred = model.predict()
    assert len(pred) == 300 and min(pred) == 0
    model.cluster_method = "bogus"
    with pytest.raises(ValueError):
        model.predict()


def test_simple_gcdec_matches_reference_golden(cuda_device, heads):
    from dance_amd.modules.spatial.spatial_domain.spagcn import SimpleGCDEC
    h = heads
    m = SimpleGCDEC(12, 12, device="cuda").to(cuda_device)
    t = lambda a: torch.from_numpy(a).to(cuda_device)
    m.gc.weight.data, m.gc.bias.data = t(h["gcdec_w"]), t(h["gcdec_b"])
    m.mu = torch.nn.Parameter(t(h["gcdec_mu"]))
    z, q = m.forward(t(h["gcdec_x"]), t(h["gcdec_adj"]))
    p = m.target_distribution(q)
    loss = m.loss_function(p.data, q)
    loss.backward()
    assert rel_err(z.detach().cpu().numpy(), h["gcdec_z"]) < TOL
    assert rel_err(q.detach().cpu().numpy(), h["gcdec_q"]) < TOL
    assert rel_err(p.detach().cpu().numpy(), h["gcdec_p"]) < TOL
    assert abs(loss.item() - float(h["gcdec_loss"])) < 1e-5 * max(1, abs(float(h["gcdec_loss"])))
    assert rel_err(m.gc.weight.grad.cpu().numpy(), h["gcdec_dw"]) < 1e-3  # tiny gradients of a KL near its optimum
    assert rel_err(m.mu.grad.cpu().numpy(), h["gcdec_dmu"]) < 1e-3


def test_spagcn_calculate_p_and_search_l(cuda_device):
    from dance_amd.modules.spatial.spatial_domain.spagcn import SpaGCN, calculate_p
    rng = np.random.default_rng(0)
    xyz = (rng.random((500, 3)) * 40).astype(np.float32)
    adj = om.pairwise_distance(xyz, 0)
    for l in (0.5, 2.0, 10.0):
        assert abs(calculate_p(adj, l) - osp.calculate_p(adj, l)) < 1e-4 * max(1.0, osp.calculate_p(adj, l))
    model = SpaGCN(device="cuda")
    l_hip, l_ref = model.search_l(0.5, adj, start=0.01, end=1000, tol=0.01, max_run=100), osp.search_l(0.5, adj)
    assert l_ref is not None and l_hip == pytest.approx(l_ref, rel=1e-6)  # same bisection path
    model.set_l(l_hip)
    e = model.calc_adj_exp(adj).cpu().numpy()
    assert rel_err(e, osp.calc_adj_exp(adj, l_hip)) < 1e-5


def test_spagcn_fit_predict_dense_and_truncated(cuda_device):
    """Planted spatial domains; dense-adjacency SpaGCN (reference arithmetic) and the kNN-truncated CSR variant
     must both recover them and agree with each other."""
    from dance_amd import kernels
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.spatial.spatial_domain.spagcn import SpaGCN, refine
    from sklearn.metrics import adjusted_rand_score
    torch.manual_seed(0)
    rng = np.random.default_rng(1)
    side = 24
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    n = xy.shape[0]
    dom = (xy[:, 0] >= side / 2).astype(int) + 2 * (xy[:, 1] >= side / 2).astype(int)
    embed = (np.eye(4)[dom] @ rng.standard_normal((4, 10)) * 2 + rng.standard_normal((n, 10))).astype(np.float32)
    adj = om.pairwise_distance(xy, 0)
    model = SpaGCN(device="cuda")
    l = model.search_l(0.5, adj)
    model.set_l(l)
    np.random.seed(0)
    model.fit((embed, adj), init="kmeans", n_clusters=4, epochs=60, lr=0.01, tol=1e-4)
    pred = model.predict((embed, adj))
    assert adjusted_rand_score(dom, pred) > 0.9
    prob = model.predict_proba((embed, adj))
    assert prob.shape == (n, 4) and torch.allclose(prob.sum(1), torch.ones(n, device=prob.device), atol=1e-5)
    refined = refine(list(range(n)), pred, adj, shape="square")
    assert adjusted_rand_score(dom, refined) >= adjusted_rand_score(dom, pred) - 1e-9
    # truncated variant: distances to the 40 nearest spots as a CSR graph; the Gaussian tail beyond them is < 1e-3
    idx, dist = kernels.knn(torch.from_numpy(xy).cuda(), 40)
    rowptr = torch.arange(0, n * 40 + 1, 40, dtype=torch.int32, device="cuda")
    order = torch.argsort(idx, dim=1)
    g = CSRGraph(rowptr, torch.gather(idx, 1, order).reshape(-1).contiguous(), torch.gather(dist, 1, order).reshape(-1).contiguous(), n, n)
    sparse = SpaGCN(l, device="cuda")
    np.random.seed(0)
    sparse.fit((embed, g), init="kmeans", n_clusters=4, epochs=60, lr=0.01, tol=1e-4)
    assert adjusted_rand_score(pred, sparse.predict((embed, g))) > 0.99
    with pytest.raises(ValueError):
        SpaGCN(device="cuda").fit((embed, adj))  # l must be set first (spagcn.py:855-856)
    # the reference's DEFAULT init ("louvain" = neighbours + leiden, spagcn.py:480-492) must run: GPU neighbour graph +
    # host modularity clustering; the number of clusters comes out of the partition
    default = SpaGCN(l, device="cuda")
    default.fit((embed, adj), epochs=60, lr=0.01, tol=1e-4, res=0.4)
    assert default.model.n_clusters == len(np.unique(default.model.trajectory[0])) >= 2
    assert adjusted_rand_score(dom, default.predict((embed, adj))) > 0.8


def test_scdsc_model_matches_reference_golden(cuda_device, heads):
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSCModel
    h = heads
    kw = json.loads(str(h["scdsc_kw"]))
    model = ScDSCModel(**kw, device="cuda").eval()
    sd = {k.split("::", 1)[1]: torch.from_numpy(v) for k, v in h.items() if k.startswith("scdsc_sd::")}
    assert set(sd) == set(model.state_dict())  # reference checkpoints load unchanged
    model.load_state_dict(sd)
    n = h["scdsc_x"].shape[0]
    a = sp.csr_matrix((h["scdsc_adj_data"], h["scdsc_adj_indices"], h["scdsc_adj_indptr"]), shape=(n, n)).tocoo()
    adj = torch.sparse_coo_tensor(np.vstack((a.row, a.col)).astype(np.int64), a.data, (n, n)).to(cuda_device)
    with torch.no_grad():
        x_bar, q, predict, z3, _mean, _disp, _pi, zinb = model(torch.from_numpy(h["scdsc_x"]).to(cuda_device), adj)
    for name, val in (("x_bar", x_bar), ("q", q), ("predict", predict), ("z3", z3), ("mean", _mean), ("disp", _disp), ("pi", _pi)):
        assert rel_err(val.cpu().numpy(), h[f"scdsc_{name}"]) < TOL, name
    loss = zinb(torch.rand(n, kw["n_input"], device=cuda_device).round(), _mean, _disp, _pi, torch.ones(n, device=cuda_device))
    assert torch.isfinite(loss)


def test_gc_dec_vs_reference_gpu(cuda_device):
    """SpaGCN's two-layer DEC model (spagcn.py:588-697) on the kernels vs the reference's own class (tests/golden/gc_dec.npz)."""
    import test_models_host_logic as mh
    mh.check_gc_dec("cuda")
