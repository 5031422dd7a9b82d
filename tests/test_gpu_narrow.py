"""GPU: the fused narrow GCN layer (dh_gcn_narrow_forward_f32 / _backward_f32, csrc/gcn_narrow.hip) — SpaGCN's GraphConvolution
50 -> 50 of BASELINE config 5 — against the float64 restatement of spagcn.py:357-363 (spmm(adj, mm(x, W)) + b) and its autograd,
through the kernels and through the layer classes (golden of the reference's own GraphConvolution included)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _graph(n, k, seed, dev, ragged=True):
    from dance_amd.graph import CSRGraph
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 2 * k, n) if ragged else np.full(n, k)
    deg[:3] = (0, 1, 70)  # an empty row, a single edge, more edges than lanes in a group
    rows = np.repeat(np.arange(n), deg)
    cols = rng.integers(0, n, rows.size)
    a = sp.csr_matrix((rng.uniform(0.1, 1.0, rows.size).astype(np.float32), (rows, cols)), shape=(n, n))
    a.sum_duplicates()
    a.sort_indices()
    return CSRGraph.from_scipy(a, dev), a


@pytest.mark.parametrize("fin,fout,ld", [(50, 50, 50), (50, 50, 52), (3, 64, 4), (63, 1, 63 + 1), (32, 10, 32), (17, 33, 18)])
def test_narrow_kernels_vs_float64(cuda_device, fin, fout, ld):
    from dance_amd import kernels
    n = 5000
    g, a = _graph(n, 12, fin * 100 + fout, cuda_device)
    gen = torch.Generator(device="cpu").manual_seed(0)
    xbuf = torch.full((n, ld), float("nan"))        # padding columns hold NaN: they must never reach the result
    xbuf[:, :fin] = torch.randn(n, fin, generator=gen)
    x = xbuf.to(cuda_device)[:, :fin]
    w = (torch.randn(fin, fout, generator=gen) / fin**0.5).to(cuda_device)
    b = torch.randn(fout, generator=gen).to(cuda_device)
    dy = torch.randn(n, fout, generator=gen).to(cuda_device)
    a64 = torch.from_numpy(a.toarray()).double().to(cuda_device)
    for act in (kernels.ACT_NONE, kernels.ACT_RELU):
        y, agg = kernels.gcn_narrow_forward(g.rowptr, g.col, g.val, x, w, b, act)
        ref_agg = a64 @ x.double()
        ref = ref_agg @ w.double() + b.double()
        ref = torch.relu(ref) if act else ref
        assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 1e-5
        assert rel_err(agg[:, :fin].cpu().numpy(), ref_agg.cpu().numpy()) < 1e-5
        assert bool((agg[:, fin:63] == 0).all()) and bool((agg[:, 63] == 1).all())
        gmask = dy.double() if not act else torch.where(ref > 0, dy.double(), torch.zeros_like(ref))
        dw, db = kernels.gcn_narrow_backward(agg, dy, fin, y_act=y if act else None)
        assert rel_err(dw.cpu().numpy(), (ref_agg.t() @ gmask).cpu().numpy()) < 2e-5
        assert rel_err(db.cpu().numpy(), gmask.sum(0).cpu().numpy()) < 2e-5
        dw2, db2 = kernels.gcn_narrow_backward(agg, dy, fin, y_act=y if act else None)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)  # fixed reduction order
        y_nb, none = kernels.gcn_narrow_forward(g.rowptr, g.col, g.val, x, w, None, act, want_agg=False)
        assert none is None
    assert not kernels.gcn_narrow_supported(64, 10) and not kernels.gcn_narrow_supported(10, 65) and kernels.gcn_narrow_supported(63, 64)


def test_graph_convolution_layer_uses_fused_path_and_matches_reference_golden(cuda_device, golden_gcn):
    """The reference's own GraphConvolution (tests/golden/gcn_layers.npz: 40 -> 24 with bias on a sparse adjacency) through the fused
    path (forced on at this tiny size) and through the generic chain: outputs, dW, db, dX."""
    from dance_amd import autograd, kernels
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.spatial.spatial_domain.spagcn import GraphConvolution
    gd = golden_gcn
    n = gd["x"].shape[0]
    adj = sp.csr_matrix((gd["adj_data"], gd["adj_indices"], gd["adj_indptr"]), shape=(n, n))
    graph = CSRGraph.from_scipy(adj, cuda_device)
    for fused, min_rows in ((True, 1), (False, 1 << 30)):
        old = autograd.NARROW_MIN_ROWS
        autograd.NARROW_MIN_ROWS = min_rows
        try:
            layer = GraphConvolution(gd["x"].shape[1], gd["w"].shape[1]).to(cuda_device)
            layer.weight.data, layer.bias.data = torch.from_numpy(gd["w"]).to(cuda_device), torch.from_numpy(gd["b"]).to(cuda_device)
            x = torch.from_numpy(gd["x"]).to(cuda_device).requires_grad_(True)
            with kernels.KernelTimer() as t:
                y = layer(x, graph)
                y.backward(torch.from_numpy(gd["dy"]).to(cuda_device))
                torch.cuda.synchronize()
            names = set(t.summary())
            assert ("gcn_narrow_forward_f32" in names) == fused and ("gcn_narrow_backward_f32" in names) == fused
            assert rel_err(y.detach().cpu().numpy(), gd["gc_sparse_out"]) < 1e-4
            assert rel_err(layer.weight.grad.cpu().numpy(), gd["gc_sparse_dW"]) < 1e-4
            assert rel_err(layer.bias.grad.cpu().numpy(), gd["gc_sparse_db"]) < 1e-4
            assert rel_err(x.grad.cpu().numpy(), gd["gc_sparse_dX"]) < 1e-4
        finally:
            autograd.NARROW_MIN_ROWS = old


def test_narrow_layer_full_size_500k(cuda_device):
    """config 5 at size: 500k spots, spatial k = 15 graph, 50 -> 50 with bias; sampled rows of Y and all of dW / db vs float64."""
    from dance_amd import kernels
    from dance_amd.autograd import gcn_layer
    from dance_amd.graph import CSRGraph
    n, k, f = 500_000, 15, 50
    gen = torch.Generator(device=cuda_device).manual_seed(5)
    col = torch.randint(0, n, (n, k), device=cuda_device, generator=gen).sort(dim=1).values.to(torch.int32).reshape(-1)
    rowptr = torch.arange(0, n * k + 1, k, dtype=torch.int32, device=cuda_device)
    val = torch.rand(n * k, device=cuda_device, generator=gen)
    graph = CSRGraph(rowptr, col, val, n, n)
    x = torch.randn(n, f, device=cuda_device, generator=gen)
    w = (torch.randn(f, f, device=cuda_device, generator=gen) / 7).requires_grad_(True)
    b = torch.randn(f, device=cuda_device, generator=gen).requires_grad_(True)
    dy = torch.randn(n, f, device=cuda_device, generator=gen)
    with kernels.KernelTimer() as t:
        y = gcn_layer(x, w, graph, b, False)
        y.backward(dy)
        torch.cuda.synchronize()
    assert set(t.summary()) == {"gcn_narrow_forward_f32", "gcn_narrow_backward_f32"}
    rows = torch.from_numpy(np.random.default_rng(0).choice(n, 200, replace=False)).to(cuda_device)
    nb = col.reshape(n, k)[rows].long()
    agg_rows = (val.reshape(n, k)[rows].double()[:, :, None] * x.double()[nb]).sum(1)
    assert rel_err(y[rows].detach().cpu().numpy(), (agg_rows @ w.detach().double() + b.detach().double()).cpu().numpy()) < 1e-5
    agg = torch.zeros(n, f, dtype=torch.float64, device=cuda_device)
    for j in range(k):  # dense float64 A X without the edge-list tensor: k gathers
        agg += val.reshape(n, k)[:, j].double()[:, None] * x.double()[col.reshape(n, k)[:, j].long()]
    assert rel_err(w.grad.cpu().numpy(), (agg.t() @ dy.double()).cpu().numpy()) < 1e-5
    assert rel_err(b.grad.cpu().numpy(), dy.double().sum(0).cpu().numpy()) < 1e-5
