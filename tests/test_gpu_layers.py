"""GPU parity of the mirrored layer classes (dance_amd.modules...) against (i) golden vectors produced by the
reference's own classes and (ii) the CPU oracle on larger seeded inputs; fp32 within 1e-4 max-norm relative."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err
from oracle import layers as ol

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _adj(g):
    n = g["x"].shape[0]
    return sp.csr_matrix((g["adj_data"], g["adj_indices"], g["adj_indptr"]), shape=(n, n))


@pytest.mark.parametrize("active,tag", [(True, "gnn_act"), (False, "gnn_lin")])
@pytest.mark.parametrize("adj_kind", ["torch_coo", "csrgraph"])
def test_gnnlayer_golden(cuda_device, golden_gcn, active, tag, adj_kind):
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.single_modality.clustering.scdsc import GNNLayer
    g = golden_gcn
    layer = GNNLayer(g["x"].shape[1], g["w"].shape[1]).to(cuda_device)
    layer.weight.data = torch.from_numpy(g["w"].copy()).to(cuda_device)
    adj = ol.scipy_to_torch_coo(_adj(g)).to(cuda_device) if adj_kind == "torch_coo" else CSRGraph.from_scipy(_adj(g), cuda_device)
    x = torch.from_numpy(g["x"].copy()).to(cuda_device).requires_grad_(True)
    y = layer(x, adj, active=active)
    y.backward(torch.from_numpy(g["dy"]).to(cuda_device))
    assert rel_err(y.detach().cpu().numpy(), g[f"{tag}_out"]) < TOL
    assert rel_err(layer.weight.grad.cpu().numpy(), g[f"{tag}_dW"]) < TOL
    assert rel_err(x.grad.cpu().numpy(), g[f"{tag}_dX"]) < TOL


@pytest.mark.parametrize("tag", ["gc_sparse", "gc_dense"])
def test_graphconvolution_golden(cuda_device, golden_gcn, tag):
    from dance_amd.modules.spatial.spatial_domain.spagcn import GraphConvolution
    g = golden_gcn
    layer = GraphConvolution(g["x"].shape[1], g["w"].shape[1]).to(cuda_device)
    assert repr(layer) == f"GraphConvolution({g['x'].shape[1]} -> {g['w'].shape[1]})"
    layer.weight.data = torch.from_numpy(g["w"].copy()).to(cuda_device)
    layer.bias.data = torch.from_numpy(g["b"].copy()).to(cuda_device)
    adj = (ol.scipy_to_torch_coo(_adj(g)) if tag == "gc_sparse" else torch.from_numpy(g["adj_dense"])).to(cuda_device)
    x = torch.from_numpy(g["x"].copy()).to(cuda_device).requires_grad_(True)
    y = layer(x, adj)
    y.backward(torch.from_numpy(g["dy"]).to(cuda_device))
    assert rel_err(y.detach().cpu().numpy(), g[f"{tag}_out"]) < TOL
    assert rel_err(layer.weight.grad.cpu().numpy(), g[f"{tag}_dW"]) < TOL
    assert rel_err(layer.bias.grad.cpu().numpy(), g[f"{tag}_db"]) < TOL
    assert rel_err(x.grad.cpu().numpy(), g[f"{tag}_dX"]) < TOL


@pytest.mark.parametrize("n,fin,fout,k", [(3000, 2000, 512, 15), (20000, 200, 50, 15), (5000, 50, 50, 6)])
def test_gnnlayer_vs_oracle_medium(cuda_device, n, fin, fout, k):
    """Seeded rand-k graph (SURVEY.md §8d 'rand-k15': k distinct random in-neighbours, values 1/k)."""
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.single_modality.clustering.scdsc import GNNLayer
    rng = np.random.default_rng(n + fin)
    x = rng.standard_normal((n, fin)).astype(np.float32)
    w = (rng.standard_normal((fin, fout)) / np.sqrt(fin)).astype(np.float32)
    dy = rng.standard_normal((n, fout)).astype(np.float32)
    cols = np.stack([rng.choice(n, k, replace=False) for _ in range(n)])
    cols.sort(axis=1)
    adj = sp.csr_matrix((np.full(n * k, 1.0 / k, np.float32), cols.ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))
    ref = ol.gcn_layer_fwd_bwd(x, adj, w, dy, active=True, x_requires_grad=True)
    layer = GNNLayer(fin, fout).to(cuda_device)
    layer.weight.data = torch.from_numpy(w).to(cuda_device)
    xt = torch.from_numpy(x).to(cuda_device).requires_grad_(True)
    y = layer(xt, CSRGraph.from_scipy(adj, cuda_device))
    y.backward(torch.from_numpy(dy).to(cuda_device))
    y_hip = y.detach().cpu().numpy()
    assert rel_err(y_hip, ref["out"]) < TOL
    # ReLU is not differentiable at 0: an output that is +-1e-8 flips its mask under fp32 re-association and
    # moves dW by one whole dy entry.  Such flips are legal only where the oracle's pre-activation is ~0 ...
    flips = (y_hip > 0) != (ref["out"] > 0)
    scale = np.abs(ref["out"]).max()
    assert flips.sum() <= 8 and np.all(np.abs(ref["out"][flips]) + np.abs(y_hip[flips]) < 1e-5 * scale)
    # ... and the gradients must match the oracle arithmetic evaluated (in float64) with the HIP mask.
    g = np.where(y_hip > 0, dy, 0).astype(np.float64)
    ds = adj.T.astype(np.float64) @ g
    assert rel_err(layer.weight.grad.cpu().numpy(), x.astype(np.float64).T @ ds) < TOL
    assert rel_err(xt.grad.cpu().numpy(), ds @ w.astype(np.float64).T) < TOL
    if not flips.any():
        assert rel_err(layer.weight.grad.cpu().numpy(), ref["dW"]) < TOL
        assert rel_err(xt.grad.cpu().numpy(), ref["dX"]) < TOL


def test_same_sparse_adj_every_epoch(cuda_device, golden_gcn):
    """The reference loop hands the SAME torch sparse adjacency to the layer every epoch (scdsc.py:257-288): the cached
    CSR must be found again (the second call used to raise) and give identical results."""
    from dance_amd import graph
    from dance_amd.modules.single_modality.clustering.scdsc import GNNLayer
    from dance_amd.modules.spatial.spatial_domain.spagcn import GraphConvolution
    g = golden_gcn
    adj = ol.scipy_to_torch_coo(_adj(g)).to(cuda_device)
    x = torch.from_numpy(g["x"].copy()).to(cuda_device)
    layer = GNNLayer(g["x"].shape[1], g["w"].shape[1]).to(cuda_device)
    outs = [layer(x, adj).detach().clone() for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert graph.as_graph(adj) is graph.as_graph(adj)
    gc_layer = GraphConvolution(g["x"].shape[1], g["w"].shape[1]).to(cuda_device)
    assert torch.equal(gc_layer(x, adj), gc_layer(x, adj))


def test_premasked_backward_variant_is_bit_identical(cuda_device, monkeypatch):
    """DANCE_AMD_BWD_MASK=premask (dy masked once, then the plain SpMM — an A/B switch, not the default) computes the same
    sums in the same order as the fused-mask backward: dW and dX bit for bit."""
    from dance_amd import autograd
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.single_modality.clustering.scdsc import GNNLayer
    rng = np.random.default_rng(5)
    n, fin, fout, k = 4000, 96, 512, 7
    x = torch.from_numpy(rng.standard_normal((n, fin)).astype(np.float32)).to(cuda_device)
    dy = torch.from_numpy(rng.standard_normal((n, fout)).astype(np.float32)).to(cuda_device)
    cols = np.stack([rng.choice(n, k, replace=False) for _ in range(n)])
    cols.sort(axis=1)
    adj = sp.csr_matrix((rng.random(n * k).astype(np.float32), cols.ravel(), np.arange(0, n * k + 1, k)), shape=(n, n))
    graph = CSRGraph.from_scipy(adj, cuda_device)
    torch.manual_seed(0)
    layer = GNNLayer(fin, fout).to(cuda_device)
    grads = {}
    for mode in ("fused", "premask"):
        monkeypatch.setattr(autograd, "BWD_MASK_MODE", mode)
        xt = x.clone().requires_grad_(True)
        layer.weight.grad = None
        layer(xt, graph).backward(dy)
        grads[mode] = (layer.weight.grad.clone(), xt.grad.clone())
    assert torch.equal(grads["fused"][0], grads["premask"][0]) and torch.equal(grads["fused"][1], grads["premask"][1])
