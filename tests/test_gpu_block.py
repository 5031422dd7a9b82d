"""GPU: the device block builder (dh_block_plan / dh_block_fill) against the torch-op restatement of the full-neighbour
sampler semantics (tests/cpu_ops.block_build) — bit-exact row pointers, local columns, values and source order."""
import numpy as np
import pytest
import torch

import cpu_ops

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(n, avg_deg, seed):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, n)
    deg[rng.integers(0, n, max(n // 50, 1))] = 0            # isolated rows
    rowptr = np.concatenate(([0], np.cumsum(deg))).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(n, d, replace=False)) for d in deg] + [np.empty(0, np.int64)]).astype(np.int32)
    val = rng.random(col.size).astype(np.float32)
    return torch.from_numpy(rowptr).to(DEV), torch.from_numpy(col).to(DEV), torch.from_numpy(val).to(DEV)


@pytest.mark.parametrize("n,avg_deg,n_seeds", [(5000, 12, 700), (2049, 3, 1), (4096, 40, 4096), (300, 5, 17), (100_000, 20, 30_000)])
def test_block_build_matches_restatement(cuda_device, n, avg_deg, n_seeds):
    from dance_amd import kernels
    rowptr, col, val = _graph(n, avg_deg, n + n_seeds)
    mark = torch.zeros(n, dtype=torch.uint8, device=DEV)
    lut = torch.empty(n, dtype=torch.int32, device=DEV)
    gen = torch.Generator().manual_seed(n_seeds)
    for rep in range(3):  # the scratch arrays are reused: the mark array must come back all-zero
        seeds = torch.randperm(n, generator=gen)[:n_seeds].to(DEV)
        got = kernels.block_build(rowptr, col, val, seeds, mark, lut)
        ref = cpu_ops.block_build(rowptr, col, val, seeds, None, None)
        for a, b, name in zip(got, ref, ("rowptr", "col", "val", "src_ids")):
            assert a.dtype == b.dtype and torch.equal(a, b), (name, rep)
        assert int(mark.sum()) == 0
    got = kernels.block_build(rowptr, col, None, seeds, mark, lut)     # structure only
    assert got[2] is None and torch.equal(got[1], ref[1])


def test_block_build_empty_and_sampler(cuda_device):
    from dance_amd import kernels
    from dance_amd.cellgraph import CellGeneGraph, NeighborSampler
    rowptr, col, val = _graph(1000, 8, 1)
    mark, lut = torch.zeros(1000, dtype=torch.uint8, device=DEV), torch.empty(1000, dtype=torch.int32, device=DEV)
    brp, bcol, bval, src = kernels.block_build(rowptr, col, val, torch.empty(0, dtype=torch.int64, device=DEV), mark, lut)
    assert brp.tolist() == [0] and bcol.numel() == 0 and src.numel() == 0
    g = CellGeneGraph(rowptr, col, val, None, 1000, {"features": torch.arange(1000., device=DEV)[:, None]})
    seeds = torch.tensor([5, 900, 17], device=DEV)
    inp, out, blocks = NeighborSampler([-1, -1]).sample(g, seeds)      # two layers: the inner block's seeds are the outer block's sources
    assert torch.equal(out, seeds) and torch.equal(blocks[1].dstdata["_ID"], seeds)
    assert torch.equal(blocks[0].dstdata["_ID"], blocks[1].srcdata["_ID"]) and torch.equal(inp, blocks[0].srcdata["_ID"])
    assert torch.equal(blocks[1].srcdata["features"][:, 0].long(), blocks[1].srcdata["_ID"])


def test_degree_scales_kernel(cuda_device):
    """dh_csr_degree_scales_f32 against the torch formulation it replaced (in-degree from the row pointer, out-degree by scatter-add,
    clamp at 1, pow(-0.5) / reciprocal) and against float64; the entries behind row n_rows (a static block's padding tail) are not counted."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    from dance_amd import kernels
    g = torch.Generator().manual_seed(0)
    n_rows, n_cols = 777, 300
    deg = torch.randint(0, 9, (n_rows, ), generator=g)
    deg[5] = 0
    rowptr = torch.zeros(n_rows + 2, dtype=torch.int32)
    rowptr[1:n_rows + 1] = torch.cumsum(deg, 0).to(torch.int32)
    nnz = int(rowptr[n_rows])
    rowptr[n_rows + 1] = nnz + 50                                    # a padding row over 50 tail entries
    col = torch.randint(0, n_cols, (nnz + 50, ), generator=g).to(torch.int32)
    col[nnz:] = 0
    for mode, pad in ((kernels.DEGREE_BOTH, 1), (kernels.DEGREE_BOTH, 0), (kernels.DEGREE_MEAN, 1)):
        want_r, want_c = cpu_ops.degree_scales(rowptr, col, n_rows, n_cols, mode, n_pad=pad)
        got_r, got_c = kernels.degree_scales(rowptr.to(cuda_device), col.to(cuda_device), n_rows, n_cols, mode, n_pad=pad)
        assert got_r.shape == (n_rows + pad, ) and float((got_r.cpu() - want_r).abs().max()) <= 1e-7
        if mode == kernels.DEGREE_BOTH:
            # 1 / sqrt(d) with correctly rounded operations vs pow(d, -0.5) on the host: one rounding apart at most
            assert float((got_c.cpu() / want_c - 1).abs().max()) <= 2e-7
            exact = (torch.bincount(col[:nnz].long(), minlength=n_cols).double().clamp(min=1)**-0.5)
            assert float((got_c.cpu().double() / exact - 1).abs().max()) <= 1.2e-7
        else:
            assert got_c is None
    r0, c0 = kernels.degree_scales(torch.zeros(4, dtype=torch.int32, device=cuda_device), torch.zeros(0, dtype=torch.int32, device=cuda_device), 3, 5)
    assert torch.equal(r0, torch.ones(3, device=cuda_device)) and torch.equal(c0, torch.ones(5, device=cuda_device))


def test_static_cell_block_flags_rows_without_exactly_one_self_loop(cuda_device):
    """The captured GraphSC / ScDeepSort steps take the decoder target among a batch's own cells to be the identity: every seed
    row must hold exactly one self loop.  dh_block_cells_static sets ``bad`` otherwise (a CellFeatureGraph-layout
    graph WITHOUT cell self loops used to train on a wrong target silently)."""
    from dance_amd.cellgraph import CellGeneGraph, StaticCellBlock
    n_genes, n_cells = 5, 6

    def build(self_loops):
        rows = [[] for _ in range(n_genes)]  # gene rows: empty (not used as seeds)
        for c in range(n_cells):
            r = [c % n_genes, (c + 2) % n_genes]
            rows.append(sorted(r) + ([n_genes + c] * self_loops[c]))
        rowptr = torch.tensor(np.concatenate(([0], np.cumsum([len(r) for r in rows]))), dtype=torch.int32, device=DEV)
        col = torch.tensor([x for r in rows for x in r], dtype=torch.int32, device=DEV)
        cid = torch.tensor(list(range(n_genes)) + [-1] * n_cells, dtype=torch.int32, device=DEV)
        return CellGeneGraph(rowptr, col, torch.ones(col.numel(), device=DEV), None, n_genes + n_cells,
                             {"cell_id": cid, "features": torch.zeros(n_genes + n_cells, 2, device=DEV)})

    # bit 2 = a seed without exactly one self loop (graph-sc's identity target needs it, scDeepSort's step does not: ADVICE r4);
    # bit 1 (a seed that is not a cell of the layout) stays clear for all three graphs
    for loops, expect in (([1] * 6, 0), ([1, 1, 0, 1, 1, 1], 2), ([1, 2, 1, 1, 1, 1], 2)):
        sb = StaticCellBlock(build(loops), 4)
        sb.bad.zero_()
        sb.seeds.copy_(torch.tensor([5, 6, 7, 8], device=DEV))  # cells 0 .. 3
        sb.rebuild()
        assert int(sb.bad) == expect, loops


def test_fanout_sampled_blocks_on_device(cuda_device):
    """NeighborSampler with positive fan-outs on the device builder: per destination min(fanout, degree) distinct in-edges that exist
    in the graph with their weights; two layers chain (the first block's destinations are the second's sources); -1 entries give the
    full-neighbour block; AdaptiveSAGE / WeightedGraphConv run on the sampled blocks."""
    from dance_amd.cellgraph import CellGeneGraph, NeighborSampler
    rowptr, col, val = _graph(20_000, 30, 7)
    n = 20_000
    g = CellGeneGraph(rowptr, col, val, None, n, {"features": torch.randn(n, 16, device=DEV)})
    seeds = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:3000].to(DEV)
    gen = torch.Generator(device=DEV).manual_seed(2)
    inp, out, blocks = NeighborSampler([7, 4], generator=gen).sample(g, seeds)
    assert len(blocks) == 2 and torch.equal(out, seeds)
    last, first = blocks[1], blocks[0]
    rp = rowptr.long()
    deg = (rp[seeds + 1] - rp[seeds])
    assert torch.equal((last.rowptr[1:3001] - last.rowptr[:3000]).long(), deg.clamp(max=4))
    ids = last.srcdata["_ID"]
    dst = torch.repeat_interleave(seeds, (last.rowptr[1:3001] - last.rowptr[:3000]).long())
    src = ids[last.col.long()]
    eid = last.edata["_ID"]
    assert torch.equal(col[eid].long(), src) and torch.equal(val[eid], last.val)
    owner = torch.searchsorted(rp, eid, right=True) - 1
    assert torch.equal(owner, dst)                                      # every kept edge lies in its destination's row of the graph
    key = dst * n + src
    assert key.unique().numel() == key.numel()                          # without replacement
    assert torch.equal(first.srcdata["_ID"][:first.number_of_dst_nodes()], ids) and torch.equal(inp, first.srcdata["_ID"])
    assert int((first.rowptr[1:first.number_of_dst_nodes() + 1] - first.rowptr[:first.number_of_dst_nodes()]).max()) <= 7
    full = NeighborSampler([-1]).sample(g, seeds)[2][0]
    big = NeighborSampler([10_000], generator=gen).sample(g, seeds)[2][0]
    assert torch.equal(full.rowptr, big.rowptr) and torch.equal(full.col, big.col) and torch.equal(full.val, big.val)
    # a GraphConv-style layer runs on the sampled block
    from dance_amd import autograd
    from dance_amd.graph import CSRGraph
    w = torch.randn(16, 8, device=DEV, requires_grad=True)
    blk_g = CSRGraph(last.rowptr, last.col, last.val, last.number_of_dst_nodes(), last.number_of_src_nodes())
    y = autograd.gcn_layer(last.srcdata["features"], w, blk_g, None, True)
    y.sum().backward()
    assert y.shape == (3000, 8) and bool(torch.isfinite(w.grad).all())


def test_static_block_transpose_ignores_the_padding_tail(cuda_device):
    """the transposed copy of a StaticCellBlock (built over its real rows only) must apply A^T to a gradient whose last real
    row is NON-FINITE without leaking NaN through the zero-weight padding entries — they point at the block's padding row, not at the last
    real row — and equal the transpose of the same block without its padding."""
    from dance_amd import kernels
    from dance_amd.cellgraph import StaticCellBlock
    from dance_amd.graph import CSRGraph
    from test_gpu_ministep import _graph
    g = _graph(120, 30, 8, 3, cuda_device)
    sb = StaticCellBlock(g, 16)
    sb.seeds.copy_(torch.arange(30, 46, device=cuda_device))
    sb.rebuild()
    assert int(sb.bad) == 0
    nnz = int(sb.rowptr[16])
    assert nnz < sb.e_max  # there IS a padding tail
    gr = CSRGraph(sb.rowptr, sb.col, sb.val, 17, sb.number_of_src_nodes())
    gr.t_rows = 16
    t = gr.transpose()
    dy = torch.randn(17, 5, device=cuda_device)
    dy[16] = 0.0           # the padding row of an upstream gradient is zero ...
    dy[15] = float("inf")  # ... and the last real row may be anything
    dx = kernels.spmm_csr(t.rowptr, t.col, t.val, dy, n_cols=17)
    # reference: the block's real entries only
    ref = CSRGraph(sb.rowptr[:17].clone(), sb.col[:nnz].clone(), sb.val[:nnz].clone(), 16, sb.number_of_src_nodes()).transpose()
    dx_ref = kernels.spmm_csr(ref.rowptr, ref.col, ref.val, dy[:16].contiguous(), n_cols=16)
    touched = torch.isinf(dx_ref).any(1)  # source rows with an edge into row 15
    assert touched.any() and not torch.isnan(dx).any()
    assert torch.equal(torch.isinf(dx).any(1), touched)
    fin = ~touched
    assert torch.allclose(dx[fin], dx_ref[fin], rtol=1e-6, atol=1e-6)
