"""Derived-graph caches are keyed by tensor identity + in-place version (a data_ptr/shape key returned a
stale CSR after ``edge_index.copy_(...)`` and could alias a recycled allocation)."""
import gc

import torch

from dance_amd.graph import TensorKeyedCache
from dance_amd.modules.spatial.spatial_domain.stagate import _EDGE_CACHE, edge_index_graph


def test_edge_index_graph_sees_inplace_edit():
    a = torch.tensor([[0, 1, 2, 0], [1, 2, 0, 2]])
    b = torch.tensor([[2, 2, 1, 0], [0, 1, 0, 1]])
    g1, _ = edge_index_graph(a, 3)
    assert edge_index_graph(a, 3)[0] is g1  # hit: same tensor, same version
    before = g1.col.clone()
    a.copy_(b)  # same storage, same shape: the old key could not tell
    g2, slot = edge_index_graph(a, 3)
    assert g2 is not g1
    assert not torch.equal(g2.col, before)
    src = b[0][torch.argsort(b[1] * 3 + b[0], stable=True)]
    assert torch.equal(g2.col.long(), src)
    assert torch.equal(g2.col.long()[slot], b[0])


def test_distinct_tensors_with_equal_content_do_not_alias_and_entries_die_with_the_tensor():
    cache = TensorKeyedCache()
    t = torch.arange(6).reshape(2, 3)
    u = t.clone()
    cache.put(t, "T", 3)
    assert cache.get(t, 3) == "T" and cache.get(u, 3) is None and cache.get(t, 4) is None
    n = len(cache)
    del t
    gc.collect()
    assert len(cache) == n - 1


def test_node_count_is_part_of_the_key():
    a = torch.tensor([[0, 1], [1, 0]])
    g2, _ = edge_index_graph(a, 2)
    g5, _ = edge_index_graph(a, 5)
    assert g2.n_rows == 2 and g5.n_rows == 5 and g5.rowptr.numel() == 6
    assert len(_EDGE_CACHE) >= 1


def test_permute_keeps_row_order_and_renumbers_the_cached_transpose():
    import numpy as np
    import scipy.sparse as sp
    from dance_amd.graph import CSRGraph, locality_order
    rng = np.random.default_rng(0)
    n = 40
    a = sp.random(n, n, density=0.15, random_state=1, format="csr", dtype=np.float32)
    a.sort_indices()
    g = CSRGraph.from_scipy(a, "cpu")
    at = a.T.tocsr()
    at.sort_indices()
    g._t = CSRGraph.from_scipy(at, "cpu")
    g._t._t = g
    perm = torch.from_numpy(rng.permutation(n))
    h = g.permute(perm)
    p = perm.numpy()
    want = a[p][:, p].tocsr()
    assert np.array_equal(h.to_scipy().toarray(), want.toarray())
    assert np.array_equal(h.transpose().to_scipy().toarray(), want.T.toarray())
    assert h.transpose().transpose() is h
    inv = np.argsort(p)
    for i in range(n):  # edges of new row i = edges of old row perm[i], same order, columns relabelled
        lo, hi = a.indptr[p[i]], a.indptr[p[i] + 1]
        assert np.array_equal(h.col[h.rowptr[i]:h.rowptr[i + 1]].numpy(), inv[a.indices[lo:hi]])
        assert np.array_equal(h.val[h.rowptr[i]:h.rowptr[i + 1]].numpy(), a.data[lo:hi])
    order = locality_order(g)
    assert sorted(order.tolist()) == list(range(n))
    band = lambda m: max(abs(i - j) for i, j in zip(*m.nonzero()))
    ring = sp.diags([1, 1], [1, -1], shape=(n, n), format="csr", dtype=np.float32)
    shuffled = ring[p][:, p].tocsr()
    gs = CSRGraph.from_scipy(shuffled, "cpu", symmetric=True)
    assert band(gs.permute(locality_order(gs)).to_scipy()) <= 2 < band(shuffled)


def test_from_scipy_takes_the_device_graph_of_a_lazy_obsp_slot():
    """``obsp["NeighborGraph"]`` of an on-device pipeline is a LazyScipyCSR; a model that calls ``CSRGraph.from_scipy`` on it (ScDSC.fit)
    gets the device graph itself — no host copy, no re-upload."""
    import numpy as np
    import torch
    from dance_amd.graph import CSRGraph, LazyScipyCSR
    g = CSRGraph(torch.tensor([0, 1, 3], dtype=torch.int32), torch.tensor([1, 0, 1], dtype=torch.int32), torch.tensor([1.0, 2.0, 3.0]), 2, 2)
    lazy = LazyScipyCSR(g)
    c0 = LazyScipyCSR.host_copies
    assert CSRGraph.from_scipy(lazy, "cpu") is g and LazyScipyCSR.host_copies == c0
    assert np.array_equal(lazy.toarray(), [[0, 1], [2, 3]]) and LazyScipyCSR.host_copies == c0 + 1


def test_lazy_wrappers_survive_copy_and_pickle():
    """``copy.deepcopy(data)`` / pickling a Data object whose slots hold the lazy device wrappers (a LazyScipyCSR used to recurse
    without end in ``__getattr__``)."""
    import copy
    import pickle
    import numpy as np
    import torch
    from dance_amd.data import DeviceArray
    from dance_amd.graph import CSRGraph, LazyScipyCSR
    g = CSRGraph(torch.tensor([0, 1, 3], dtype=torch.int32), torch.tensor([1, 0, 1], dtype=torch.int32), torch.tensor([1.0, 2.0, 3.0]), 2, 2)
    for clone in (copy.deepcopy, lambda o: pickle.loads(pickle.dumps(o))):
        lz = clone(LazyScipyCSR(g))
        assert isinstance(lz, LazyScipyCSR) and lz.shape == (2, 2) and np.array_equal(lz.toarray(), [[0, 1], [2, 3]])
        da = clone(DeviceArray(torch.arange(6.0).reshape(2, 3)))
        assert isinstance(da, DeviceArray) and np.array_equal(np.asarray(da), np.arange(6.0).reshape(2, 3))
    with __import__("pytest").raises(AttributeError):
        LazyScipyCSR(g)._not_there


def test_device_array_behaves_like_an_ndarray_for_host_code():
    import numpy as np
    import torch
    from dance_amd.data import DeviceArray
    a = np.arange(12.0, dtype=np.float32).reshape(3, 4)
    d, e = DeviceArray(torch.from_numpy(a.copy())), DeviceArray(torch.ones(3, 4))
    c0 = DeviceArray.host_copies
    assert np.array_equal(d * 2, a * 2) and np.array_equal(2 * d, a * 2) and np.array_equal(d + e, a + 1) and np.array_equal(1 - d, 1 - a)
    assert np.allclose(d / d.sum(1, keepdims=True), a / a.sum(1, keepdims=True)) and bool((d == d).all()) and int((d > 3).sum()) == 8
    assert np.array_equal(-d, -a) and np.array_equal(abs(d), a) and (d @ np.ones((4, 2), dtype=np.float32)).shape == (3, 2) and np.array_equal(d**2, a**2)
    assert [r.shape for r in d] == [(4, )] * 3 and hash(d) == hash(d)
    assert DeviceArray.host_copies == c0 + 2  # one materialisation per array, however many expressions


def test_get_feature_return_types_on_lazy_graph_slot():
    """Every return type of Data.get_feature on an obsp slot that holds a LazyScipyCSR ("sparse" and the device types used to fail)."""
    import numpy as np
    import scipy.sparse as sp
    import torch
    from dance_amd.data import AnnDataLite, Data, DeviceArray
    from dance_amd.graph import CSRGraph, LazyScipyCSR
    g = CSRGraph(torch.tensor([0, 1, 2, 3, 4, 5], dtype=torch.int32), torch.tensor([1, 0, 3, 2, 4], dtype=torch.int32), torch.ones(5), 5, 5)
    d = Data(AnnDataLite(DeviceArray(torch.zeros(5, 4)), obsp={"G": LazyScipyCSR(g)}), train_size=3, val_size=0, test_size=2)
    want = np.array([[0, 1, 0], [1, 0, 0], [0, 0, 0]], dtype=np.float32)
    kw = dict(split_name="train", channel="G", channel_type="obsp")
    assert np.array_equal(d.get_feature(return_type="numpy", **kw), want) and torch.equal(d.get_feature(return_type="torch", **kw), torch.from_numpy(want))
    s = d.get_feature(return_type="sparse", **kw)
    assert sp.issparse(s) and np.array_equal(s.toarray(), want) and torch.equal(d.get_feature(return_type="cpu", **kw), torch.from_numpy(want))
    assert isinstance(d.get_feature(return_type="default", channel="G", channel_type="obsp"), LazyScipyCSR)


def test_cell_filter_after_a_lazy_graph_slot():
    """filter_by_mask on a Data object whose obsp holds a LazyScipyCSR (cells filtered after an on-device graph transform)."""
    import numpy as np
    import torch
    from dance_amd.data import AnnDataLite, Data, DeviceArray
    from dance_amd.graph import CSRGraph, LazyScipyCSR
    g = CSRGraph(torch.tensor([0, 1, 2, 3, 4], dtype=torch.int32), torch.tensor([1, 0, 3, 2], dtype=torch.int32), torch.tensor([1.0, 2.0, 3.0, 4.0]), 4, 4)
    d = Data(AnnDataLite(DeviceArray(torch.arange(8.0).reshape(4, 2)), obsp={"G": LazyScipyCSR(g)}, uns={"G.hip": g, "other": 1}), train_size="all")
    d.filter_by_mask(np.array([True, False, True, True]))
    assert "G.hip" not in d.data.uns and d.data.uns["other"] == 1   # the device graph over the old cells is dropped, not left stale
    assert d.data.X.shape == (3, 2) and np.array_equal(d.data.obsp["G"].toarray(), [[0, 0, 0], [0, 0, 3], [0, 4, 0]])


def _tiny_cell_gene_graph(device="cpu"):
    """3 genes (nodes 0..2, cell_id >= 0) + 4 cells (nodes 3..6): self loops only — enough for the loader's layout test."""
    from dance_amd.cellgraph import CellGeneGraph
    n = 7
    rowptr = torch.arange(n + 1, dtype=torch.int32, device=device)
    col = torch.arange(n, dtype=torch.int32, device=device)
    cid = torch.tensor([0, 1, 2, -1, -1, -1, -1], dtype=torch.int32, device=device)
    return CellGeneGraph(rowptr, col, torch.ones(n, device=device), None, n, {"cell_id": cid, "features": torch.zeros(n, 2, device=device)})


def test_loader_cells_only_flag_is_not_cached_across_seed_sets():
    """the flag was cached under (data_ptr, numel) of the loader's private device copy; a later loader whose
    seeds had the same length but included gene nodes could hit the stale entry and get cells_only=True (wrong block layout for
    the matrix-core aggregation).  Host seeds are now reduced on the host; device seeds are keyed by identity + version."""
    from dance_amd.cellgraph import DataLoader, NeighborSampler
    g = _tiny_cell_gene_graph()
    sampler = NeighborSampler([-1])
    cells = [3, 4, 5, 6]
    for _ in range(3):  # same-sized seed sets back to back, freed in between: the allocator may recycle the block
        assert DataLoader(g, cells, sampler, batch_size=2).cells_only is True
        assert DataLoader(g, [0, 4, 5, 6], sampler, batch_size=2).cells_only is False
    t = torch.tensor(cells)
    assert DataLoader(g, t, sampler, batch_size=2).cells_only is True
    t[0] = 1  # in-place edit of the caller's tensor: a gene among the seeds now
    assert DataLoader(g, t, sampler, batch_size=2).cells_only is False
