"""GPU: the LDS-staged cell <- gene aggregation (dh_sage_aggregate_cells, sage_lds.hip) against the generic gather kernel
(dh_sage_aggregate_f32/bf16, itself pinned to the reference's AdaptiveSAGE.message_func / fn.mean golden) and against a
float64 restatement of gnn.py:62-90, in both layouts it is used with: the full cell-gene graph (genes are the first G
source rows) and a sampled block (seed cells first, genes after them, self loop last in every row)."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bipartite(n_cells, n_genes, density, seed, layout):
    """CSR rows = cells; returns rowptr, col, w, src_cell_id, dst_cell_id, gene_begin, n_src."""
    rng = np.random.default_rng(seed)
    x = (rng.random((n_cells, n_genes)) < density)
    x[rng.integers(0, n_cells, 3)] = False          # cells with no gene edge at all
    x[0] = True                                     # a cell expressing every gene
    x[1] = False                                    # ... and a fully isolated one (no gene edge, no self loop)
    gene_begin = 0 if layout == "graph" else n_cells
    self_col = (n_genes + np.arange(n_cells)) if layout == "graph" else np.arange(n_cells)
    has_self = rng.random(n_cells) < 0.9            # a few cells without self loop; one fully isolated
    has_self[1] = False
    deg = x.sum(1)
    rows, cols = np.nonzero(x)
    rowptr = np.zeros(n_cells + 1, np.int64)
    rowptr[1:] = np.cumsum(deg + has_self)
    col = np.empty(rowptr[-1], np.int32)
    pos = rowptr[:-1].copy()
    start = np.concatenate(([0], np.cumsum(deg)))
    for i in range(n_cells):
        g = cols[start[i]:start[i + 1]] + gene_begin
        col[pos[i]:pos[i] + len(g)] = g
        if has_self[i]:
            col[pos[i] + len(g)] = self_col[i]
    w = (rng.random(col.size) + 0.25).astype(np.float32)
    n_src = n_genes + n_cells
    cid = -np.ones(n_src, np.int32)
    cid[gene_begin:gene_begin + n_genes] = rng.permutation(n_genes)  # alpha index of a gene row is its cell_id, not its position
    dst_cid = -np.ones(n_cells, np.int32)
    return rowptr.astype(np.int32), col, w, cid, dst_cid, gene_begin, n_src


def _reference(rowptr, col, w, cid, alpha, h, n_genes):
    out = np.zeros((rowptr.size - 1, h.shape[1]))
    for i in range(rowptr.size - 1):
        s, t = rowptr[i], rowptr[i + 1]
        if t > s:
            c = col[s:t]
            a = np.where(cid[c] >= 0, alpha[np.maximum(cid[c], 0)], alpha[n_genes + 1])
            out[i] = ((a * w[s:t])[:, None] * h[c].astype(np.float64)).sum(0) / (t - s)
    return out


@pytest.mark.parametrize("layout", ["graph", "block"])
@pytest.mark.parametrize("n_cells,n_genes,width", [(700, 90, 400), (300, 1200, 100), (2100, 500, 200), (130, 37, 64), (64, 2000, 402)])
def test_sage_cells_f32(cuda_device, layout, n_cells, n_genes, width):
    from dance_amd import kernels
    rowptr, col, w, cid, dst_cid, gene_begin, n_src = _bipartite(n_cells, n_genes, 0.12, n_cells + width, layout)
    rng = np.random.default_rng(1)
    h = rng.standard_normal((n_src, width)).astype(np.float32)
    alpha = (rng.random(n_genes + 2) + 0.5).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    args = (t(rowptr), t(col), t(w), t(cid), t(dst_cid), t(alpha), t(h))
    got, ws = kernels.sage_aggregate_cells(*args, gene_begin, n_genes)
    old = kernels.sage_aggregate(*args)
    ref = _reference(rowptr, col, w, cid, alpha.astype(np.float64), h, n_genes)
    assert rel_err(got.cpu().numpy(), ref) < 1e-5
    assert rel_err(got.cpu().numpy(), old.cpu().numpy()) < 2e-6      # same sums, alpha folded into h instead of into w
    again, _ = kernels.sage_aggregate_cells(*args, gene_begin, n_genes, workspace=ws, reuse_segments=True)
    assert torch.equal(got, again)                                   # bit-reproducible; cached segment table


@pytest.mark.parametrize("layout", ["graph", "block"])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_sage_cells_bf16(cuda_device, layout, out_dtype):
    from dance_amd import kernels
    n_cells, n_genes, width = 1500, 900, 400
    rowptr, col, w, cid, dst_cid, gene_begin, n_src = _bipartite(n_cells, n_genes, 0.1, 5, layout)
    rng = np.random.default_rng(2)
    h16 = torch.from_numpy(rng.standard_normal((n_src, width)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    alpha = (rng.random(n_genes + 2) + 0.5).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    args = (t(rowptr), t(col), t(w), t(cid), t(dst_cid), t(alpha), h16)
    got, _ = kernels.sage_aggregate_cells(*args, gene_begin, n_genes, out_dtype=out_dtype)
    ref = _reference(rowptr, col, w, cid, alpha.astype(np.float64), h16.float().cpu().numpy(), n_genes)
    assert got.dtype == out_dtype
    if out_dtype == torch.float32:
        assert rel_err(got.cpu().numpy(), ref) < 1e-5               # exact fp32 sums of the bf16 inputs
    else:
        assert rel_err(got.float().cpu().numpy(), ref) < 4e-3       # one round-to-nearest-even on store (2^-9)
        old = kernels.sage_aggregate_bf16(*args)
        assert rel_err(got.float().cpu().numpy(), old.float().cpu().numpy()) < 8e-3


def test_sage_cells_argument_errors(cuda_device):
    from dance_amd import _lib, kernels
    rowptr, col, w, cid, dst_cid, gene_begin, n_src = _bipartite(50, 20, 0.2, 0, "graph")
    t = lambda a: torch.from_numpy(a).to(DEV)
    h = torch.zeros(n_src, 63, device=DEV)
    alpha = torch.ones(22, device=DEV)
    with pytest.raises(_lib.DanceHipError, match="even width"):
        kernels.sage_aggregate_cells(t(rowptr), t(col), t(w), t(cid), t(dst_cid), alpha, h, 0, 20)
    h = torch.zeros(n_src, 64, device=DEV)
    with pytest.raises(_lib.DanceHipError, match="outside"):
        kernels.sage_aggregate_cells(t(rowptr), t(col), t(w), t(cid), t(dst_cid), alpha, h, 60, 20)
