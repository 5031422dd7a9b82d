"""CPU: the DGL graph-object surface the reference's layers and builders call on OUR graph / block objects —
``update_all`` with a message UDF and with the built-in pairs, ``adjacency_matrix``, ``add_edges``.

The strongest pin available without dgl: the reference's OWN ``AdaptiveSAGE`` (dance/models/nn/gnn.py:8-96, lifted from
/root/reference by AST) runs here against ``dance_amd.cellgraph.Block`` — its ``forward`` drives ``block.local_scope()``,
``block.srcdata / dstdata``, ``number_of_dst_nodes()`` and ``update_all(self.message_func, dgl.function.mean("m", "neigh"))`` —
and must produce the ``neigh`` of our fused kernels' restatement (oracle.sage) and the output of ``dance_amd.nn.AdaptiveSAGE``."""
import logging
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

import cpu_ops
from conftest import rel_err
from oracle import ref_extract, sage as osage


@pytest.fixture
def cpu_kernels(monkeypatch):
    from dance_amd import kernels
    for name in cpu_ops.STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    return kernels


def _cell_gene_graph(n_cells=40, n_genes=12, d=6, seed=0):
    """CellFeatureGraph layout: genes 0..G-1, cells after; cell <-> gene edges both ways + self loops, weights > 0."""
    from dance_amd.cellgraph import CellGeneGraph
    rng = np.random.default_rng(seed)
    x = (rng.random((n_cells, n_genes)) < 0.3) * rng.random((n_cells, n_genes))
    ci, gi = np.nonzero(x)
    src = np.concatenate((ci + n_genes, gi, np.arange(n_cells + n_genes)))
    dst = np.concatenate((gi, ci + n_genes, np.arange(n_cells + n_genes)))
    w = np.concatenate((x[ci, gi], x[ci, gi], np.ones(n_cells + n_genes))).astype(np.float32)
    n = n_cells + n_genes
    order = np.lexsort((np.arange(len(src)), dst))
    rowptr = np.concatenate(([0], np.cumsum(np.bincount(dst, minlength=n)))).astype(np.int32)
    cid = np.concatenate((np.arange(n_genes), -np.ones(n_cells))).astype(np.int32)
    feats = rng.standard_normal((n, d)).astype(np.float32)
    g = CellGeneGraph(torch.from_numpy(rowptr), torch.from_numpy(src[order].astype(np.int32)), torch.from_numpy(w[order]),
                      torch.from_numpy(order.astype(np.int32)), n, {"cell_id": torch.from_numpy(cid), "features": torch.from_numpy(feats)})
    return g, (src, dst, w, cid, feats)


@pytest.mark.skipif(not ref_extract.available(), reason="needs /root/reference (build container only)")
def test_reference_adaptive_sage_runs_on_our_block(cpu_kernels):
    from dance_amd import function as fn
    from dance_amd.cellgraph import NeighborSampler
    from dance_amd.nn import AdaptiveSAGE as OurSAGE
    g, (src, dst, w, cid, feats) = _cell_gene_graph()
    n_genes, d, hid = 12, 6, 5
    dgl_ns = types.SimpleNamespace(function=fn)
    RefSAGE = ref_extract.extract("dance/models/nn/gnn.py", "AdaptiveSAGE", {"dgl": dgl_ns, "logger": logging.getLogger("reference")})
    alpha = nn.Parameter(torch.from_numpy((np.random.default_rng(1).random((n_genes + 2, 1)) + 0.5).astype(np.float32)))
    torch.manual_seed(0)
    ref_layer = RefSAGE(d, hid, alpha, nn.Identity(), nn.ReLU(), nn.Identity())
    ours = OurSAGE(d, hid, alpha, nn.Identity(), nn.ReLU(), nn.Identity())
    ours.load_state_dict(ref_layer.state_dict())
    seeds = torch.tensor([12 + 3, 12 + 17, 12 + 0, 12 + 39, 12 + 8])
    _, _, blocks = NeighborSampler([-1]).sample(g, seeds)
    blk = blocks[0]
    h = blk.srcdata["features"]
    z_ref = ref_layer(blk, h)                       # the reference's forward, on OUR block object
    neigh_ref = blk.dstdata["neigh"]                # what its update_all left behind (gnn.py:90: computed, then unused)
    # the same block's edges as explicit lists -> the oracle restatement of message_func + fn.mean
    ids = blk.srcdata["_ID"].numpy()
    rp, col, val = blk.rowptr.numpy(), blk.col.numpy(), blk.val.numpy()
    e_dst = np.repeat(np.arange(len(seeds)), np.diff(rp))
    want = osage.sage_neigh(col, e_dst, val, cid[ids], cid[ids[:len(seeds)]], alpha.detach().numpy(), feats[ids], len(seeds))
    assert rel_err(neigh_ref.detach().numpy(), want) < 1e-6
    z_ours = ours(blk, h)
    assert torch.allclose(z_ref, z_ours, rtol=1e-6, atol=1e-7)
    assert rel_err(ours.last_neigh.detach().numpy(), want) < 1e-6
    # the whole graph as one "block" (srcdata = dstdata = ndata): the reference's message function over every edge
    g.ndata["h"] = g.ndata["features"]
    g.update_all(ref_layer.message_func, fn.mean("m", "neigh"))
    full = osage.sage_neigh(src, dst, w, cid, cid, alpha.detach().numpy(), feats, g.number_of_nodes())
    assert rel_err(g.ndata["neigh"].detach().numpy(), full) < 1e-6


def test_builtin_messages_and_adjacency(cpu_kernels):
    from dance_amd import function as fn
    from dance_amd.cellgraph import NeighborSampler
    g, (src, dst, w, cid, feats) = _cell_gene_graph(seed=3)
    n = g.number_of_nodes()
    g.ndata["h"] = g.ndata["features"]
    dense = np.zeros((n, n), np.float64)
    np.add.at(dense, (dst, src), w)
    g.update_all(fn.u_mul_e("h", "weight", "m"), fn.sum("m", "s"))
    assert rel_err(g.ndata["s"].numpy(), dense @ feats) < 1e-6
    g.update_all(fn.copy_u("h", "m"), fn.mean("m", "mu"))
    ones = (dense > 0).astype(np.float64)
    assert rel_err(g.ndata["mu"].numpy(), (ones @ feats) / np.maximum(ones.sum(1), 1)[:, None]) < 1e-6
    # graphsc.py:463: a UDF h_src * w_e with fn.sum equals the built-in pair
    g.update_all(lambda edges: {"m": edges.src["h"] * edges.data["weight"]}, fn.sum("m", "s2"))
    assert torch.allclose(g.ndata["s2"], g.ndata["s"], rtol=1e-6, atol=1e-7)
    a = g.adjacency_matrix().to_dense().numpy()     # rows = sources (DGL 1.x), one per stored edge
    assert a.shape == (n, n) and np.array_equal(a, ones.T)
    assert np.array_equal(g.adjacency_matrix(transpose=True).to_dense().numpy(), ones)
    seeds = torch.tensor([14, 20, 33])
    blk = NeighborSampler([-1]).sample(g, seeds)[2][0]
    ab = blk.adjacency_matrix().to_dense()          # graphsc.py:208-209: [num_src, num_dst], then the dst x dst corner
    assert ab.shape == (blk.number_of_src_nodes(), 3)
    corner = ab[blk.dstnodes()][:, blk.dstnodes()]
    assert torch.equal(corner, torch.eye(3))        # among a batch's own cells only the self loops
    with pytest.raises(KeyError):
        g.update_all(fn.copy_u("h", "m"), fn.sum("other", "x"))
    with pytest.raises(TypeError):
        g.update_all(fn.copy_u("h", "m"), lambda nodes: {})


def test_add_edges_appends_in_edge_id_order(cpu_kernels):
    """cell_feature_graph.py:69: ``g.add_edges(g.nodes(), g.nodes(), {"weight": ones[:, None]})`` after the weighted edges."""
    from dance_amd.cellgraph import CellGeneGraph
    src = torch.tensor([2, 0, 1, 2]); dst = torch.tensor([0, 1, 2, 1])
    order = torch.argsort(dst * 10 + torch.arange(4))
    rowptr = torch.tensor([0, 1, 3, 4], dtype=torch.int32)
    g = CellGeneGraph(rowptr, src[order].to(torch.int32), torch.tensor([.5, 1.5, 2.5, 3.5])[order], order.to(torch.int32), 3)
    g.add_edges(g.nodes(), g.nodes(), {"weight": torch.ones(3)[:, None]})
    s, d = g.edges()
    assert s.tolist() == [2, 0, 1, 2, 0, 1, 2] and d.tolist() == [0, 1, 2, 1, 0, 1, 2]   # old edges first, then the new ones in order
    assert g.edata["weight"].reshape(-1).tolist() == [.5, 1.5, 2.5, 3.5, 1., 1., 1.]
    assert g.in_degrees().tolist() == [2, 3, 2] and g.number_of_edges() == 7
    s1, d1, e1 = g.in_edges(1, form="all")
    assert s1.tolist() == [0, 2, 1] and e1.tolist() == [1, 3, 5]
    g.add_edges(torch.tensor([0]), torch.tensor([2]))          # no data: DGL zero-fills
    assert g.edata["weight"].reshape(-1)[-1].item() == 0.0 and g.number_of_edges() == 8
    with pytest.raises(ValueError):
        g.add_edges(torch.tensor([5]), torch.tensor([0]))
    with pytest.raises(KeyError):
        g.add_edges(torch.tensor([0]), torch.tensor([0]), {"colour": torch.ones(1)})
