"""CPU (kernels replaced by the torch stand-ins of tests/cpu_ops.py): ``cellgraph.NeighborSampler`` with positive fan-outs —
dgl.dataloading.NeighborSampler(fanouts, edge_dir="in") as the reference constructs it (scdeepsort.py:183 passes ``[-1] * n_layers``;
any other list is a legal argument of the same class): at most ``fanout`` in-edges per destination, uniformly without replacement,
kept edges carry the graph's weights, -1 entries are the full-neighbour blocks, the last entry applies to the hop next to the seeds."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import cpu_ops


@pytest.fixture()
def cpu_kernels(monkeypatch):
    from dance_amd import kernels
    for name in cpu_ops.STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    return kernels


def _graph(n_cells=60, n_genes=25, seed=0):
    from dance_amd.cellgraph import CellGeneGraph
    rng = np.random.default_rng(seed)
    x = (rng.random((n_cells, n_genes)) < 0.4)
    x[np.arange(n_cells), rng.integers(0, n_genes, n_cells)] = True
    n = n_genes + n_cells
    a = sp.lil_matrix((n, n), dtype=np.float32)  # a[dst, src]
    for c, gsel in enumerate(x):
        for gi in np.nonzero(gsel)[0]:
            w = rng.uniform(0.1, 1.0)
            a[n_genes + c, gi] = w
            a[gi, n_genes + c] = w
    for i in range(n):
        a[i, i] = 1.0
    a = a.tocsr()
    a.sort_indices()
    cid = np.concatenate((np.arange(n_genes), -np.ones(n_cells))).astype(np.int32)
    g = CellGeneGraph(torch.from_numpy(a.indptr.astype(np.int32)), torch.from_numpy(a.indices.astype(np.int32)), torch.from_numpy(a.data.astype(np.float32)),
                      None, n, {"cell_id": torch.from_numpy(cid), "features": torch.randn(n, 4)})
    return g, a, n_genes


def _block_edges(blk):
    """(dst node id, src node id, weight) of every block edge, in the parent graph's numbering."""
    ids = blk.srcdata["_ID"].numpy()
    rp, col, val = blk.rowptr.numpy(), blk.col.numpy(), blk.val.numpy()
    dst = np.repeat(ids[:blk.number_of_dst_nodes()], np.diff(rp[:blk.number_of_dst_nodes() + 1]))
    return dst, ids[col], val


def test_fanout_blocks_are_subsets_with_the_graphs_weights(cpu_kernels):
    from dance_amd.cellgraph import NeighborSampler
    g, a, n_genes = _graph()
    seeds = torch.tensor([n_genes + i for i in (7, 0, 33, 12, 59)])
    gen = torch.Generator().manual_seed(0)
    for k in (1, 3, 8, 1000):
        _, out_nodes, blocks = NeighborSampler([k], generator=gen).sample(g, seeds)
        blk = blocks[0]
        assert torch.equal(out_nodes, seeds) and torch.equal(blk.srcdata["_ID"][:5], seeds)     # dst nodes first (dgl.to_block)
        dst, src, w = _block_edges(blk)
        for s in seeds.tolist():
            sel = dst == s
            deg = a.indptr[s + 1] - a.indptr[s]
            assert sel.sum() == min(k, deg)
            assert len(set(src[sel])) == sel.sum()                                                  # without replacement
            for u, ww in zip(src[sel], w[sel]):
                assert a[s, u] == ww                                                                # an edge of the graph, with its weight
        # "_SLOT" points at those very entries of the parent CSR; "_ID" holds the parent's EDGE ids of the same entries (DGL's
        # meaning: on a CellFeatureGraph the CSR slots are not in edge order)
        slot = blk.edata["_SLOT"].numpy()
        assert np.array_equal(g.col.numpy()[slot], src) and np.array_equal(g.val.numpy()[slot], w)
        want = slot if g.eid is None else g.eid.numpy()[slot]
        assert np.array_equal(blk.edata["_ID"].numpy(), want)
        if g.eid is not None:   # edge-id-ordered views of the parent give the same edges
            e_src, e_dst = g.edges()
            assert np.array_equal(e_src.numpy()[want], src)
    full = NeighborSampler([-1]).sample(g, seeds)[2][0]
    big = NeighborSampler([1000], generator=gen).sample(g, seeds)[2][0]
    assert torch.equal(full.rowptr, big.rowptr) and torch.equal(full.col, big.col) and torch.equal(full.val, big.val)


def test_fanout_order_and_layers(cpu_kernels):
    """fanouts = [f0, f1]: DGL walks them reversed — f1 for the hop next to the seeds (the LAST block), f0 for the first block."""
    from dance_amd.cellgraph import NeighborSampler
    g, a, n_genes = _graph(seed=3)
    seeds = torch.arange(n_genes, n_genes + 10)
    inp, _, blocks = NeighborSampler([2, 5], generator=torch.Generator().manual_seed(1)).sample(g, seeds)
    assert len(blocks) == 2 and torch.equal(blocks[1].srcdata["_ID"], blocks[0].srcdata["_ID"][:blocks[0].number_of_dst_nodes()])
    assert torch.equal(inp, blocks[0].srcdata["_ID"])
    deg_last = np.diff(blocks[1].rowptr.numpy()[:11])
    deg_first = np.diff(blocks[0].rowptr.numpy()[:blocks[0].number_of_dst_nodes() + 1])
    assert deg_last.max() <= 5 and deg_first.max() <= 2 and (deg_last == 5).any() and (deg_first == 2).any()
    with pytest.raises(ValueError):
        NeighborSampler([0])
    with pytest.raises(NotImplementedError):
        NeighborSampler([-1], edge_dir="out")


def test_fanout_sampling_is_uniform(cpu_kernels):
    """Every in-edge of a node of degree d is kept with probability k / d: counts over 3000 draws within 5 sigma."""
    from dance_amd.cellgraph import NeighborSampler
    g, a, n_genes = _graph(n_cells=40, n_genes=30, seed=5)
    s = n_genes + 3
    nbrs = a.indices[a.indptr[s]:a.indptr[s + 1]]
    d, k, trials = len(nbrs), 4, 3000
    assert d > k
    sampler = NeighborSampler([k], generator=torch.Generator().manual_seed(7))
    hits = {int(u): 0 for u in nbrs}
    for _ in range(trials):
        blk = sampler.sample(g, torch.tensor([s, n_genes + 9]))[2][0]
        _, src, _ = _block_edges(blk)
        for u in src[:k]:
            hits[int(u)] += 1
    p = k / d
    sigma = (trials * p * (1 - p))**0.5
    assert all(abs(h - trials * p) < 5 * sigma for h in hits.values()), (hits, trials * p, sigma)
