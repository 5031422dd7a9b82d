"""graph.morton_order / locality_order(method="morton"): a permutation, deterministic, and local — rows adjacent in the order are close
in the space the kNN graph was built in; a kNN graph renumbered by it has a far smaller column spread per row.  CPU tensors."""
import numpy as np
import torch

from dance_amd.graph import CSRGraph, locality_order, morton_order


def test_morton_order_is_a_local_permutation():
    g = torch.Generator().manual_seed(0)
    centres = torch.randn(8, 12, generator=g) * 6
    x = centres[torch.randint(0, 8, (4000, ), generator=g)] + torch.randn(4000, 12, generator=g)
    p = morton_order(x)
    assert sorted(p.tolist()) == list(range(4000)) and torch.equal(p, morton_order(x))
    step_ord = (x[p][1:] - x[p][:-1]).norm(dim=1).mean()
    step_in = (x[1:] - x[:-1]).norm(dim=1).mean()
    assert float(step_ord) < 0.5 * float(step_in)
    # exact kNN graph (k = 8) of the same points: the mean |row - column| shrinks by an order of magnitude after renumbering
    d2 = torch.cdist(x, x)
    idx = d2.topk(8, largest=False).indices
    inv = torch.empty(4000, dtype=torch.int64)
    inv[p] = torch.arange(4000)
    spread_in = (idx - torch.arange(4000)[:, None]).abs().float().mean()
    spread_ord = (inv[idx] - inv[torch.arange(4000)][:, None]).abs().float().mean()
    assert float(spread_ord) < 0.2 * float(spread_in)
    rowptr = torch.arange(0, 4000 * 8 + 1, 8, dtype=torch.int32)
    graph = CSRGraph(rowptr, idx.sort(dim=1).values.reshape(-1).to(torch.int32), None, 4000, 4000)
    assert torch.equal(locality_order(graph, "morton", coords=x), p)


def test_morton_order_degenerate_inputs():
    assert morton_order(torch.zeros(5, 3)).tolist() == [0, 1, 2, 3, 4]          # all equal: ties keep the input order
    assert sorted(morton_order(torch.arange(7.0)[:, None]).tolist()) == list(range(7))
    x = torch.tensor([[3.0], [1.0], [2.0]])
    assert morton_order(x).tolist() in ([1, 2, 0], [0, 2, 1])                     # along the one axis, either direction
