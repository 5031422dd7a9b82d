"""Shared body of the (f)4 "free rider" parity checks: the mirrored layers of scGNN2 / DSTG / STdGCN / GraphSCI against
tests/golden/free_riders.npz (outputs of the reference's own classes, tests/golden/make_golden.py::make_free_riders).
Run on CPU tensors with the kernel stand-ins (tests/test_free_riders.py) and on the GPU through the kernels
(tests/test_gpu_free_riders.py)."""
import os

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
import torch.nn.functional as F

from conftest import GOLDEN, rel_err

TOL = 1e-4  # fp32 layer outputs and gradients, max-norm relative


def gold():
    return np.load(os.path.join(GOLDEN, "free_riders.npz"))


def _graph(g, dev):
    from dance_amd.graph import CSRGraph
    n = g["fr_x"].shape[0]
    return CSRGraph.from_scipy(sp.csr_matrix((g["fr_adj_data"], g["fr_adj_indices"], g["fr_adj_indptr"]), shape=(n, n)), dev)


def _load(layer, g, tag):
    sd = {k.split("::", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_sd::")}
    assert sorted(sd) == sorted(layer.state_dict()), (tag, sorted(sd), sorted(layer.state_dict()))
    layer.load_state_dict(sd)


def _check_layer(g, tag, layer, dev, adj):
    _load(layer, g, tag)
    layer.to(dev)
    xt = torch.from_numpy(g["fr_x"]).to(dev).requires_grad_(True)
    y = layer(xt, adj)
    y.backward(torch.from_numpy(g["fr_dy"]).to(dev))
    assert rel_err(y.detach().cpu().numpy(), g[tag + "_out"]) < TOL, tag
    assert rel_err(xt.grad.cpu().numpy(), g[tag + "_dX"]) < TOL, tag
    for k, p in layer.named_parameters():
        assert rel_err(p.grad.cpu().numpy(), g[f"{tag}_grad::{k}"]) < TOL, (tag, k)


def check_gcn_family(dev):
    from dance_amd.modules.single_modality.imputation import scgnn2
    from dance_amd.modules.spatial.cell_type_deconvo import dstg, stdgcn
    g = gold()
    adj = _graph(g, dev)
    fin, fout = g["fr_x"].shape[1], g["fr_dy"].shape[1]
    _check_layer(g, "s2gc_relu", scgnn2.GraphConvolution(fin, fout, 0., act=F.relu), dev, adj)
    _check_layer(g, "s2gc_lin", scgnn2.GraphConvolution(fin, fout, 0., act=lambda t: t), dev, adj)
    _check_layer(g, "dstg_nobias", dstg.GraphConvolution(fin, fout, None, bias=False), dev, adj)
    _check_layer(g, "dstg_bias", dstg.GraphConvolution(fin, fout, None, bias=True), dev, adj)
    _check_layer(g, "dstg_gcn", dstg.GCN(fin, 16, fout, bias=False, dropout=0.), dev, adj)
    _check_layer(g, "stdgcn", stdgcn.conGraphConvolutionlayer(fin, fout, bias=True), dev, adj)
    # the layers also accept what the reference hands them: a torch sparse COO adjacency
    coo = adj.to_scipy().tocoo()
    t = torch.sparse_coo_tensor(np.vstack((coo.row, coo.col)).astype(np.int64), coo.data, coo.shape).to(dev)
    layer = stdgcn.conGraphConvolutionlayer(fin, fout, bias=True)
    _load(layer, g, "stdgcn")
    with torch.no_grad():
        y = layer.to(dev)(torch.from_numpy(g["fr_x"]).to(dev), t)
    assert rel_err(y.cpu().numpy(), g["stdgcn_out"]) < TOL


def check_gat(dev):
    from dance_amd.modules.single_modality.imputation import scgnn2
    g = gold()
    fin = g["fr_x"].shape[1]
    ei = torch.from_numpy(g["fr_edge_index"]).to(dev)
    for tag, kw, fo in (("gat_concat", dict(num_of_heads=2, concat=True, activation=nn.ELU()), 7),
                        ("gat_mean", dict(num_of_heads=3, concat=False, activation=None), 7),
                        ("gat_same", dict(num_of_heads=2, concat=True, activation=None), fin)):
        layer = scgnn2.GATLayer(fin, fo, dropout_prob=0.0, add_skip_connection=True, bias=True, log_attention_weights=True, **kw)
        _load(layer, g, tag)
        layer.to(dev)
        xt = torch.from_numpy(g["fr_x"]).to(dev).requires_grad_(True)
        y, ei_out = layer((xt, ei))
        assert ei_out is ei
        y.backward(torch.from_numpy(g[tag + "_dy"]).to(dev))
        assert rel_err(y.detach().cpu().numpy(), g[tag + "_out"]) < TOL, tag
        assert rel_err(layer.attention_weights.detach().cpu().numpy(), g[tag + "_att"]) < TOL, tag
        assert rel_err(xt.grad.cpu().numpy(), g[tag + "_dX"]) < 5 * TOL, tag
        for k, p in layer.named_parameters():
            key = f"{tag}_grad::{k}"
            if key in g.files:
                assert rel_err(p.grad.cpu().numpy(), g[key]) < 5 * TOL, (tag, k)
            else:  # skip_proj of the FIN == FOUT layer is never used
                assert p.grad is None, (tag, k)
    gae = scgnn2.Graph_AE(fin, 6, gat_dropout=0, multi_heads=2, gat_hid_embed=8)
    _load(gae, g, "gae")
    gae.to(dev).eval()
    with torch.no_grad():
        z, info, rec = gae(torch.from_numpy(g["fr_x"]).to(dev), ei, use_GAT=True)
        z2, info2, rec2 = gae(torch.from_numpy(g["fr_x"]).to(dev), _graph(g, dev), use_GAT=False)
    assert info is None
    assert rel_err(z.cpu().numpy(), g["gae_gat_z"]) < TOL and rel_err(rec.cpu().numpy(), g["gae_gat_recon"]) < TOL
    assert rel_err(z2.cpu().numpy(), g["gae_gcn_z"]) < TOL and rel_err(info2[1].cpu().numpy(), g["gae_gcn_logvar"]) < TOL
    assert rel_err(rec2.cpu().numpy(), g["gae_gcn_recon"]) < TOL
    # attention dropout: expectation preserved, aggregation uses the dropped coefficients
    layer = scgnn2.GATLayer(fin, 7, num_of_heads=2, dropout_prob=0.5, log_attention_weights=True).to(dev)
    layer.train()
    torch.manual_seed(0)
    y, _ = layer((torch.from_numpy(g["fr_x"]).to(dev), ei))
    a = layer.attention_weights
    assert bool((a == 0).any()) and bool((a > 0).any()) and y.shape == (g["fr_x"].shape[0], 14)


def check_graphsci(dev):
    from dance_amd.modules.single_modality.imputation import graphsci
    g = gold()
    fin = g["fr_x"].shape[1]
    gm = graphsci.GNNModel(in_feats=fin, out_feats=9, dropout=0., n_hidden1=16, n_hidden2=10)
    _load(gm, g, "gsci")
    gm.to(dev)
    graph = _graph(g, dev)  # stored values must be ignored (DGL's GraphConv without edge_weight)
    graph.ndata = {"feat": torch.from_numpy(g["fr_x"]).to(dev)}
    h = gm.conv2(graph, gm.conv1(graph, graph.ndata["feat"]))
    mean = gm.dec_mean(graph, h)
    mean.backward(torch.from_numpy(g["gsci_dy"]).to(dev))
    assert rel_err(h.detach().cpu().numpy(), g["gsci_h"]) < TOL and rel_err(mean.detach().cpu().numpy(), g["gsci_mean"]) < TOL
    for k, p in gm.named_parameters():
        key = f"gsci_grad::{k}"
        if key in g.files:
            assert rel_err(p.grad.cpu().numpy(), g[key]) < TOL, k
    with torch.no_grad():
        z_adj, log_std, z_mean = gm(graph)
    assert rel_err(z_mean.cpu().numpy(), g["gsci_fwd_mean"]) < TOL and rel_err(log_std.cpu().numpy(), g["gsci_fwd_log_std"]) < TOL
    assert z_adj.shape == z_mean.shape
    # norm variants against the closed form
    x = torch.from_numpy(g["fr_x"]).to(dev)
    a = (graph.to_scipy() != 0).astype(np.float64).toarray()
    din, dout = np.maximum(a.sum(1), 1), np.maximum(a.sum(0), 1)
    for norm, l, r in (("none", np.ones_like(din), np.ones_like(dout)), ("right", 1 / din, np.ones_like(dout)),
                       ("left", np.ones_like(din), 1 / dout), ("both", din**-0.5, dout**-0.5)):
        conv = graphsci.GraphConv(fin, 5, norm=norm).to(dev)
        want = (l[:, None] * a * r[None, :]) @ (g["fr_x"].astype(np.float64) @ conv.weight.detach().cpu().numpy().astype(np.float64))
        assert rel_err(conv(graph, x).detach().cpu().numpy(), want) < TOL, norm
