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
