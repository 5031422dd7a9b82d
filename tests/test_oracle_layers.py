"""CPU: pin the oracle restatement of the GCN layers against golden vectors produced by the reference's
own classes (tests/golden/make_golden.py), and — when the reference tree is present — against the reference
code directly."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err
from oracle import layers as ol
from oracle import ref_extract

TOL = 1e-6  # same torch-CPU ops, same order: agreement is essentially exact


def _adj(g):
    n = g["x"].shape[0]
    return sp.csr_matrix((g["adj_data"], g["adj_indices"], g["adj_indptr"]), shape=(n, n))


@pytest.mark.parametrize("active,tag", [(True, "gnn_act"), (False, "gnn_lin")])
def test_gnnlayer_matches_reference_golden(golden_gcn, active, tag):
    g = golden_gcn
    r = ol.gcn_layer_fwd_bwd(g["x"], _adj(g), g["w"], g["dy"], active=active, x_requires_grad=True)
    assert rel_err(r["out"], g[f"{tag}_out"]) < TOL
    assert rel_err(r["dW"], g[f"{tag}_dW"]) < TOL
    assert rel_err(r["dX"], g[f"{tag}_dX"]) < TOL


@pytest.mark.parametrize("tag", ["gc_sparse", "gc_dense"])
def test_graphconvolution_matches_reference_golden(golden_gcn, tag):
    g = golden_gcn
    adj = _adj(g) if tag == "gc_sparse" else g["adj_dense"]
    r = ol.gcn_layer_fwd_bwd(g["x"], adj, g["w"], g["dy"], bias=g["b"], active=False, x_requires_grad=True)
    for k in ("out", "dW", "db", "dX"):
        assert rel_err(r[k], g[f"{tag}_{k}"]) < TOL, k


def test_oracle_modules_match_functional(golden_gcn):
    g = golden_gcn
    layer = ol.GNNLayer(g["x"].shape[1], g["w"].shape[1])
    assert tuple(layer.weight.shape) == g["w"].shape and list(dict(layer.named_parameters())) == ["weight"]
    layer.weight.data = torch.from_numpy(g["w"].copy())
    y = layer(torch.from_numpy(g["x"]), ol.scipy_to_torch_coo(_adj(g)))
    assert rel_err(y.detach().numpy(), g["gnn_act_out"]) < TOL
    gc = ol.GraphConvolution(g["x"].shape[1], g["w"].shape[1])
    assert sorted(dict(gc.named_parameters())) == ["bias", "weight"]
    stdv = 1 / np.sqrt(g["w"].shape[1])
    assert float(gc.weight.abs().max()) <= stdv and float(gc.bias.abs().max()) <= stdv


@pytest.mark.skipif(not ref_extract.available(), reason="reference tree only exists in the build container")
def test_oracle_matches_live_reference_classes():
    rng = np.random.default_rng(3)
    n, fin, fout = 50, 17, 9
    x = rng.standard_normal((n, fin)).astype(np.float32)
    w = rng.standard_normal((fin, fout)).astype(np.float32)
    dy = rng.standard_normal((n, fout)).astype(np.float32)
    adj = sp.random(n, n, density=0.1, random_state=1, format="csr", dtype=np.float32)
    Ref = ref_extract.extract("dance/modules/single_modality/clustering/scdsc.py", "GNNLayer")
    layer = Ref(fin, fout)
    layer.weight.data = torch.from_numpy(w.copy())
    y = layer(torch.from_numpy(x), ol.scipy_to_torch_coo(adj))
    y.backward(torch.from_numpy(dy))
    r = ol.gcn_layer_fwd_bwd(x, adj, w, dy, active=True)
    assert rel_err(r["out"], y.detach().numpy()) < TOL
    assert rel_err(r["dW"], layer.weight.grad.numpy()) < TOL
