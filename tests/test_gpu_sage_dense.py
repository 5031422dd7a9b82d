"""GPU: the densified-operand AdaptiveSAGE aggregation (dh_csr_densify_window + dh_sage_tail + MFMA GEMM) against the
gather kernel (pinned to the reference's message_func / fn.mean golden) and a float64 restatement of gnn.py:62-90, for
cell destinations (window = the gene rows) in graph and block layout and for gene destinations (window = the cell rows)."""
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


def _ref(rowptr, col, w, cid_src, cid_dst, alpha, h, n_genes):
    out = np.zeros((rowptr.size - 1, h.shape[1]))
    for i in range(rowptr.size - 1):
        s, t = rowptr[i], rowptr[i + 1]
        if t > s:
            c = col[s:t]
            sid, did = cid_src[c], cid_dst[i]
            idx = np.full(c.size, n_genes + 1)
            idx = np.where((sid >= 0) & (did < 0), sid, idx)
            idx = np.where((did >= 0) & (sid < 0), did, idx)
            idx = np.where((did >= 0) & (sid >= 0), n_genes, idx)
            out[i] = ((alpha[idx] * w[s:t])[:, None] * h[c].astype(np.float64)).sum(0) / (t - s)
    return out


def _gene_rows(n_cells, n_genes, density, seed):
    """CSR rows = genes of the full graph (nodes: genes [0,G), cells [G,G+N)): in-edges from cells + the gene's self loop."""
    rng = np.random.default_rng(seed)
    x = rng.random((n_genes, n_cells)) < density
    x[2] = False                                    # a gene no cell expresses: only its self loop
    rowptr = np.zeros(n_genes + 1, np.int64)
    rowptr[1:] = np.cumsum(x.sum(1) + 1)
    col = np.empty(rowptr[-1], np.int32)
    for g in range(n_genes):
        cells = np.nonzero(x[g])[0] + n_genes
        col[rowptr[g]] = g                          # self loop sorts first (gene id < cell ids)
        col[rowptr[g] + 1:rowptr[g + 1]] = cells
    w = (rng.random(col.size) + 0.25).astype(np.float32)
    cid = np.concatenate((rng.permutation(n_genes), -np.ones(n_cells))).astype(np.int32)
    return rowptr.astype(np.int32), col, w, cid, cid[:n_genes].copy()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("layout", ["graph", "block"])
@pytest.mark.parametrize("n_cells,n_genes,width", [(700, 90, 400), (300, 1203, 104), (1000, 500, 200)])
def test_dense_cells(cuda_device, dtype, layout, n_cells, n_genes, width):
    from dance_amd import kernels
    rowptr, col, w, cid, dst_cid, gene_begin, n_src = _bipartite(n_cells, n_genes, 0.12, n_cells + width, layout)
    rng = np.random.default_rng(1)
    h = torch.from_numpy(rng.standard_normal((n_src, width)).astype(np.float32)).to(DEV)
    alpha = (rng.random(n_genes + 2) + 0.5).astype(np.float32)
    if dtype == "bf16":
        h = h.to(torch.bfloat16)
    t = lambda a: torch.from_numpy(a).to(DEV)
    args = (t(rowptr), t(col), t(w), t(cid), t(dst_cid), t(alpha), h)
    got = kernels.sage_aggregate_dense(*args, gene_begin, n_genes)
    ref = _ref(rowptr, col, w, cid, dst_cid, alpha.astype(np.float64), h.float().cpu().numpy(), n_genes)
    if dtype == "f32":
        assert rel_err(got.cpu().numpy(), ref) < 1e-5
        assert rel_err(got.cpu().numpy(), kernels.sage_aggregate(*args).cpu().numpy()) < 1e-5
    else:
        assert got.dtype == torch.bfloat16
        assert rel_err(got.float().cpu().numpy(), ref) < 1e-2   # bf16 adjacency entries + bf16 output (SURVEY.md §8c bf16 bar)
        f32 = kernels.sage_aggregate_dense(*args, gene_begin, n_genes, out_dtype=torch.float32)
        assert rel_err(f32.cpu().numpy(), ref) < 4e-3           # only the 2^-9 rounding of alpha * w / deg is left


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("layout", ["graph", "block"])
@pytest.mark.parametrize("n_cells,n_genes,width,density", [(700, 90, 400, 0.12), (300, 1203, 104, 0.12), (1000, 500, 200, 0.12),
                                                           (260, 2000, 400, 0.10), (130, 300, 448, 0.6),
                                                           # >= 256 row blocks: the unsplit kernel (the cases above split the gene window over
                                                           # blockIdx.y — except 90 genes, a single chunk — and sum the shares)
                                                           (33000, 400, 104, 0.02)])
def test_mfma_cells(cuda_device, dtype, layout, n_cells, n_genes, width, density):
    """dh_sage_window_mfma (adjacency densified per workgroup in LDS, bf16 hi + lo splits on the matrix cores) against the
    float64 restatement of gnn.py:62-90 and the gather kernel; the dense 0.6 case overflows the 16-entry stream prefetch."""
    from dance_amd import kernels
    rowptr, col, w, cid, dst_cid, gene_begin, n_src = _bipartite(n_cells, n_genes, density, n_cells + width, layout)
    rng = np.random.default_rng(1)
    h = torch.from_numpy(rng.standard_normal((n_src, width)).astype(np.float32)).to(DEV)
    alpha = (rng.random(n_genes + 2) + 0.5).astype(np.float32)
    if dtype == "bf16":
        h = h.to(torch.bfloat16)
    t = lambda a: torch.from_numpy(a).to(DEV)
    args = (t(rowptr), t(col), t(w), t(cid), t(dst_cid), t(alpha), h)
    ref = _ref(rowptr, col, w, cid, dst_cid, alpha.astype(np.float64), h.float().cpu().numpy(), n_genes)
    if dtype == "f32":
        got = kernels.sage_aggregate_mfma(*args, gene_begin, n_genes)
        # three exact bf16 x bf16 products per term: 2^-18 residuals of the two splits + the dropped lo * lo
        assert rel_err(got.cpu().numpy(), ref) < 1e-5
        assert rel_err(got.cpu().numpy(), kernels.sage_aggregate(*args).cpu().numpy()) < 1e-5
        assert torch.equal(got, kernels.sage_aggregate_mfma(*args, gene_begin, n_genes))  # deterministic
    else:
        f32 = kernels.sage_aggregate_mfma(*args, gene_begin, n_genes, out_dtype=torch.float32)
        assert rel_err(f32.cpu().numpy(), ref) < 1e-5           # bf16 features are exact operands; entries hi + lo
        got = kernels.sage_aggregate_mfma(*args, gene_begin, n_genes)
        assert got.dtype == torch.bfloat16 and rel_err(got.float().cpu().numpy(), ref) < 1e-2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_dense_gene_rows(cuda_device, dtype):
    from dance_amd import kernels
    n_cells, n_genes, width = 20_000, 60, 128                   # window of 20k cell columns: the wide (scatter) path
    rowptr, col, w, cid, dst_cid = _gene_rows(n_cells, n_genes, 0.1, 3)
    rng = np.random.default_rng(4)
    h = torch.from_numpy(rng.standard_normal((n_genes + n_cells, width)).astype(np.float32)).to(DEV)
    alpha = (rng.random(n_genes + 2) + 0.5).astype(np.float32)
    if dtype == "bf16":
        h = h.to(torch.bfloat16)
    t = lambda a: torch.from_numpy(a).to(DEV)
    args = (t(rowptr), t(col), t(w), t(cid), t(dst_cid), t(alpha), h)
    got = kernels.sage_aggregate_dense(*args, n_genes, n_cells, dst_are_genes=True, max_row_nnz=int(np.diff(rowptr).max()))
    ref = _ref(rowptr, col, w, cid, dst_cid, alpha.astype(np.float64), h.float().cpu().numpy(), n_genes)
    old = (kernels.sage_aggregate_bf16 if dtype == "bf16" else kernels.sage_aggregate)(*args)
    if dtype == "f32":
        assert rel_err(got.cpu().numpy(), ref) < 1e-5 and rel_err(old.cpu().numpy(), ref) < 1e-5
    else:
        assert rel_err(got.float().cpu().numpy(), ref) < 1e-2


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("n_cells,n_genes,width,density", [
    (20_000, 60, 128, 0.1),      # 10 slices, one 64-row group
    (5_000, 203, 400, 0.1),      # ragged last slice (5000 = 2 * 2048 + 904), four groups, a ragged row block, 13 column tiles
    (1_500, 130, 64, 0.3),       # a window narrower than one slice; rows of ~450 in-window entries: the pack kernel's general path
    (70_000, 33, 200, 0.05),     # more slices (35) than sets: several slices per workgroup
    (300, 3, 8, 0.5),            # tiny everything
])
def test_splitk_gene_rows(cuda_device, dtype, n_cells, n_genes, width, density):
    """dh_sage_window_splitk (gene destinations, the cells as the K dimension, no dense adjacency) against the float64 restatement of
    gnn.py:62-90 and the gather kernel; the plan is cached per graph and the result does not change from call to call."""
    from dance_amd import kernels
    rowptr, col, w, cid, dst_cid = _gene_rows(n_cells, n_genes, density, 11)
    rng = np.random.default_rng(5)
    h = torch.from_numpy(rng.standard_normal((n_genes + n_cells, width)).astype(np.float32)).to(DEV)
    alpha = (rng.random(n_genes + 2) + 0.5).astype(np.float32)
    if dtype == "bf16":
        h = h.to(torch.bfloat16)
    t = lambda a: torch.from_numpy(a).to(DEV)
    args = (t(rowptr), t(col), t(w), t(cid), t(dst_cid), t(alpha), h)
    assert kernels.sage_splitk_supported(n_genes, n_cells, width, h.dtype, col.size)
    got = kernels.sage_aggregate_splitk(*args, n_genes, n_cells)
    ref = _ref(rowptr, col, w, cid, dst_cid, alpha.astype(np.float64), h.float().cpu().numpy(), n_genes)
    assert got.dtype == h.dtype and got.shape == (n_genes, width)
    if dtype == "f32":
        old = kernels.sage_aggregate(*args)
        assert rel_err(old.cpu().numpy(), ref) < 1e-5
        assert rel_err(got.cpu().numpy(), ref) < 2e-5
    else:
        assert rel_err(got.float().cpu().numpy(), ref) < 1e-2
    again = kernels.sage_aggregate_splitk(*args, n_genes, n_cells)       # second call: the cached plan, same bits
    assert torch.equal(got, again)
    # a window that starts and ends inside the cell rows: the entries outside it are NOT part of this call's sum apart from the
    # out-of-window rule (they take the alpha of their own index) — compare with the reference on the same graph
    if n_cells >= 5000 and dtype == "f32":
        lo, n = n_genes + 777, n_cells - 1500
        part = kernels.sage_aggregate_splitk(*args, lo, n)
        assert rel_err(part.cpu().numpy(), ref) < 2e-5


def test_densify_window_direct(cuda_device):
    """Row / column scales, mean, padding and window clipping of dh_csr_densify_window against scipy."""
    import scipy.sparse as sp

    from dance_amd import kernels
    rng = np.random.default_rng(0)
    a = sp.random(300, 500, density=0.05, random_state=1, format="csr", dtype=np.float32)
    a.sort_indices()
    rs, cs = (rng.random(300) + 0.5).astype(np.float32), (rng.random(200) + 0.5).astype(np.float32)
    t = lambda x: torch.from_numpy(x).to(DEV)
    d = kernels.csr_densify_window(t(a.indptr.astype(np.int32)), t(a.indices.astype(np.int32)), t(a.data), 100, 200, rowscale=t(rs),
                                   colscale=t(cs), mean=True, ld=208)
    deg = np.maximum(np.diff(a.indptr), 1)
    ref = a.toarray()[:, 100:300] * rs[:, None] * cs[None, :] / deg[:, None]
    assert d.shape == (300, 208) and float(d[:, 200:].abs().max()) == 0.0
    assert rel_err(d[:, :200].cpu().numpy(), ref) < 1e-6
    d16 = kernels.csr_densify_window(t(a.indptr.astype(np.int32)), t(a.indices.astype(np.int32)), t(a.data), 0, 500, dtype=torch.bfloat16)
    assert torch.equal(d16, torch.from_numpy(a.toarray()).to(DEV).to(torch.bfloat16))
