"""GPU: every libdancehip kernel of the GCN path, called through the C ABI, against CPU references
(scipy / float64 numpy) on seeded inputs.  Integer outputs must be bit-exact; f32 within 1e-4 max-norm rel
(SURVEY.md §8c); tolerances are written per test."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

F32_TOL = 1e-4


def _rand_csr(n_rows, n_cols, avg_deg, seed, long_row=None, empty_rows=()):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(avg_deg, n_rows)
    if long_row is not None:
        deg[long_row[0]] = long_row[1]
    for r in empty_rows:
        deg[r] = 0
    deg = np.minimum(deg, n_cols)
    indptr = np.zeros(n_rows + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(deg)
    indices = np.concatenate([np.sort(rng.choice(n_cols, d, replace=False)) for d in deg]) if indptr[-1] else np.zeros(0, int)
    data = rng.uniform(0.1, 1.0, indptr[-1]).astype(np.float32)
    return sp.csr_matrix((data, indices.astype(np.int32), indptr.astype(np.int32)), shape=(n_rows, n_cols))


def _dev_csr(a, dev):
    return (torch.from_numpy(a.indptr.astype(np.int32)).to(dev), torch.from_numpy(a.indices.astype(np.int32)).to(dev),
            torch.from_numpy(a.data.astype(np.float32)).to(dev))


@pytest.mark.parametrize("width", [512, 256, 128, 50, 33, 7, 1, 300, 1000])
def test_spmm_widths(cuda_device, width):
    from dance_amd import kernels
    a = _rand_csr(777, 613, 9, seed=width, long_row=(5, 200), empty_rows=(0, 10, 776))
    z = np.random.default_rng(1).standard_normal((613, width)).astype(np.float32)
    rp, c, v = _dev_csr(a, cuda_device)
    y = kernels.spmm_csr(rp, c, v, torch.from_numpy(z).to(cuda_device))
    ref = (a.astype(np.float64) @ z.astype(np.float64))
    assert rel_err(y.cpu().numpy(), ref) < F32_TOL
    assert np.all(y.cpu().numpy()[[0, 10, 776]] == 0)  # empty rows -> exactly 0


@pytest.mark.parametrize("reduce", ["sum", "mean"])
@pytest.mark.parametrize("width", [64, 50])
def test_spmm_epilogue(cuda_device, reduce, width):
    from dance_amd import kernels
    rng = np.random.default_rng(11)
    a = _rand_csr(300, 200, 6, seed=5, empty_rows=(3,))
    z = rng.standard_normal((200, width)).astype(np.float32)
    bias = rng.standard_normal(width).astype(np.float32)
    rs = rng.uniform(0.5, 2, 300).astype(np.float32)
    cs = rng.uniform(0.5, 2, 200).astype(np.float32)
    rp, c, v = _dev_csr(a, cuda_device)
    t = lambda x: torch.from_numpy(x).to(cuda_device)
    y = kernels.spmm_csr(rp, c, v, t(z), rowscale=t(rs), colscale=t(cs), bias=t(bias), act=kernels.ACT_RELU,
                         reduce=kernels.REDUCE_MEAN if reduce == "mean" else kernels.REDUCE_SUM)
    agg = a.astype(np.float64) @ (cs[:, None].astype(np.float64) * z)
    if reduce == "mean":
        deg = np.diff(a.indptr)
        agg = agg / np.maximum(deg, 1)[:, None]
    ref = np.maximum(rs[:, None] * agg + bias, 0)
    assert rel_err(y.cpu().numpy(), ref) < F32_TOL


def test_spmm_unweighted_and_strided(cuda_device):
    from dance_amd import kernels
    a = _rand_csr(100, 100, 5, seed=2)
    big = torch.randn(100, 96, device=cuda_device)
    z = big[:, 16:80]  # row stride 96, width 64: leading dimension != width
    rp, c, _ = _dev_csr(a, cuda_device)
    y = kernels.spmm_csr(rp, c, None, z)
    pattern = sp.csr_matrix((np.ones_like(a.data), a.indices, a.indptr), shape=a.shape)
    assert rel_err(y.cpu().numpy(), pattern.astype(np.float64) @ z.cpu().numpy().astype(np.float64)) < F32_TOL


@pytest.mark.parametrize("shape", [(300, 200, 9), (64, 5000, 3), (5000, 64, 40), (1, 1, 1), (10, 10, 0), (129, 2128, 201), (576, 2576, 201)])
def test_csr_transpose_bit_exact(cuda_device, shape):
    from dance_amd import kernels
    n_rows, n_cols, deg = shape
    a = _rand_csr(n_rows, n_cols, deg, seed=sum(shape)) if deg else sp.csr_matrix((n_rows, n_cols), dtype=np.float32)
    rp, c, v = _dev_csr(a, cuda_device)
    rpt, ct, vt, perm = kernels.csr_transpose(rp, c, v, n_rows, n_cols)
    ref = a.tocsc()  # stable by column == CSR of A^T ordered by source row
    assert np.array_equal(rpt.cpu().numpy(), ref.indptr.astype(np.int32))
    assert np.array_equal(ct.cpu().numpy(), ref.indices.astype(np.int32))
    assert np.array_equal(vt.cpu().numpy(), ref.data.astype(np.float32))
    if a.nnz:
        assert np.array_equal(a.data[perm.cpu().numpy()], ref.data)


GEMM_SHAPES = [
    (257, 130, 100, False, False), (128, 128, 32, False, False), (1000, 512, 2000, False, False),
    (333, 50, 50, False, False),  # unaligned leading dimensions (SpaGCN 50->50)
    (200, 64, 5000, True, False),  # dW-style, split-K
    (2000, 512, 9000, True, False),
    (77, 33, 4100, True, False),
    (300, 200, 64, False, True),  # dX-style
    (129, 50, 31, False, True), (65, 70, 129, True, True), (1, 1, 1, False, False),
    # narrow-layer kernels (gemm_skinny.hip): rows form, reduce form, transposed B, N > K and N < K, edge sizes
    (5000, 50, 50, False, False), (5000, 50, 50, False, True), (50, 50, 9000, True, False), (3000, 10, 32, False, False),
    (4097, 64, 64, False, True), (64, 64, 100_001, True, False), (33, 7, 5000, True, False), (2000, 64, 3, False, False),
    (1024, 1, 64, False, False),
]


@pytest.mark.parametrize("M,N,K,ta,tb", GEMM_SHAPES)
def test_gemm_f32(cuda_device, M, N, K, ta, tb):
    from dance_amd import kernels
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
    c = kernels.gemm(torch.from_numpy(a).to(cuda_device), torch.from_numpy(b).to(cuda_device), trans_a=ta, trans_b=tb)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    assert c.shape == ref.shape
    assert rel_err(c.cpu().numpy(), ref) < 1e-5  # exact-f32 MFMA chain: f32 round-off only


@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_f32_k_boundaries(cuda_device, ta, tb):
    """The refill pipeline of gemm_f32.hip switches between fast, fast-tail and (unaligned) guarded K-steps around
    K = 32 t: every boundary, for aligned operands in both tile configurations, plus an accumulate into C."""
    from dance_amd import kernels
    g = torch.Generator(device=cuda_device).manual_seed(5)

    def check(M, N, K, accumulate=False):
        a = torch.randn((K, M) if ta else (M, K), device=cuda_device, generator=g)
        b = torch.randn((N, K) if tb else (K, N), device=cuda_device, generator=g)
        c0 = torch.randn(M, N, device=cuda_device, generator=g) if accumulate else None
        ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double() + (c0.double() if accumulate else 0)
        got = kernels.gemm(a, b, trans_a=ta, trans_b=tb, out=None if c0 is None else c0.clone(), accumulate=accumulate, mode="exact")
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (M, N, K, ta, tb, accumulate, err)

    for K in (4, 8, 28, 32, 36, 60, 64, 68, 96, 100, 132, 2000):
        check(260, 264, K)                    # 128 x 128 configuration, ragged edge tiles
    check(260, 264, 100, accumulate=True)
    for K in (36, 64, 100):
        check(4100, 8192, K)                  # 256 x 256 configuration (>= 512 tiles), ragged M
    if ta and not tb:
        for K in (8200, 65536 + 36, 300_004):  # split-K slices with a K tail in the last slice
            check(520, 512, K)


def test_gemm_transpose_detecting_and_accumulate(cuda_device):
    # asymmetric operands (A = I-like selector, B[i][j] = i*1000 + j) catch swapped C rows/cols
    from dance_amd import kernels
    n = 160
    a = torch.eye(n, device=cuda_device)
    b = (torch.arange(n, device=cuda_device)[:, None] * 1000 + torch.arange(n, device=cuda_device)[None, :]).float()
    c = kernels.gemm(a, b)
    assert torch.equal(c, b)
    c2 = kernels.gemm(a, b, out=c.clone(), accumulate=True)
    assert torch.equal(c2, 2 * b)


def test_relu_backward_and_colsum(cuda_device):
    from dance_amd import kernels
    for shape in [(1000, 512), (37, 50), (5, 3), (70_000, 50), (10_000, 17), (5000, 200), (3, 1)]:
        y = torch.randn(*shape, device=cuda_device)
        y[y.abs() < 0.1] = 0  # exact zeros: gradient must be 0 there (threshold_backward)
        dy = torch.randn(*shape, device=cuda_device)
        g = kernels.relu_backward(y, dy)
        assert torch.equal(g, torch.where(y > 0, dy, torch.zeros_like(dy)))
        s = kernels.colsum(dy)
        assert rel_err(s.cpu().numpy(), dy.cpu().numpy().astype(np.float64).sum(0)) < 1e-5


@pytest.mark.parametrize("width", [4, 52, 128, 300, 512, 2048])
def test_sddmm_csr(cuda_device, width):
    """dh_sddmm_csr_f32 == per-edge dot products (float64 reference); it is the edge-value gradient of the SpMM."""
    from dance_amd import kernels
    a = _rand_csr(300, 280, 9, seed=width, long_row=(7, 120), empty_rows=(3, ))
    rp, c, v = _dev_csr(a, cuda_device)
    g = torch.Generator().manual_seed(width)
    u, z = torch.randn(300, width, generator=g), torch.randn(280, width, generator=g)
    rows = np.repeat(np.arange(300), np.diff(a.indptr))
    ref = (u.double()[rows] * z.double()[a.indices]).sum(1).numpy()
    out = kernels.sddmm_csr(rp, c, u.to(cuda_device), z.to(cuda_device))
    assert rel_err(out.cpu().numpy(), ref) < 1e-5
    out_s = kernels.sddmm_csr(rp, c, u.to(cuda_device), z.to(cuda_device), scale=v)
    assert rel_err(out_s.cpu().numpy(), ref * a.data) < 1e-5
    # gradient identity: d/dval of sum(dY * spmm(A(val), Z)) == sddmm(dY, Z)
    val = v.clone().cpu().double().requires_grad_(True)
    dense = torch.zeros(300, 280, dtype=torch.float64)
    dense = dense.index_put((torch.from_numpy(rows), torch.from_numpy(a.indices.astype(np.int64))), val)
    (u.double() * (dense @ z.double())).sum().backward()
    assert rel_err(out.cpu().numpy(), val.grad.numpy()) < 1e-5


def test_errors_are_loud(cuda_device):
    from dance_amd import _lib, kernels
    with pytest.raises(_lib.DanceHipError):
        kernels.gemm(torch.zeros(4, 4), torch.zeros(4, 4))  # CPU tensors are refused, no fallback
    lib = _lib.load()
    assert lib.dh_spmm_csr_f32(4, 4, 4, None, None, None, None, None, None, 4, None, 4, None, 0, 0, None) < 0
    assert b"null" in lib.dh_last_error_string()
    # workspace contracts: the size queries are part of the ABI and an undersized / missing buffer is an error, not UB
    x = torch.randn(20000, 16, device=cuda_device)
    idx = torch.empty((20000, 5), dtype=torch.int32, device=cuda_device)
    dst = torch.empty((20000, 5), dtype=torch.float32, device=cuda_device)
    need = lib.dh_knn_bruteforce_f32_workspace_bytes(20000, 16, 20000, 5, kernels.KNN_FILTER)
    assert need > lib.dh_knn_bruteforce_f32_workspace_bytes(20000, 16, 20000, 5, kernels.KNN_SCAN) > 0
    small = torch.empty(need // 2, dtype=torch.uint8, device=cuda_device)
    rc = lib.dh_knn_bruteforce_f32(20000, 16, x.data_ptr(), 16, 0, 20000, 5, kernels.KNN_FILTER, idx.data_ptr(), dst.data_ptr(),
                                   small.data_ptr(), small.numel(), None)
    assert rc == -3 and b"workspace" in lib.dh_last_error_string()  # DH_ERR_WORKSPACE
    assert lib.dh_knn_bruteforce_f32(20000, 16, x.data_ptr(), 16, 0, 20000, 5, 7, idx.data_ptr(), dst.data_ptr(), small.data_ptr(),
                                     small.numel(), None) < 0 and b"algo" in lib.dh_last_error_string()
    a16 = torch.zeros((64, 64), dtype=torch.bfloat16, device=cuda_device)
    out = torch.zeros((64, 64), dtype=torch.float32, device=cuda_device)
    assert lib.dh_gemm_bf16(64, 64, 64, 1, 0, a16.data_ptr(), 64, a16.data_ptr(), 64, out.data_ptr(), 64, 0, None, 0, 0, None, 0, None) == -3
    assert lib.dh_gemm_bf16(64, 64, 64, 0, 1, a16.data_ptr(), 64, a16.data_ptr(), 64, out.data_ptr(), 64, 9, None, 0, 0, None, 0, None) < 0


@pytest.mark.parametrize("width", [128, 256, 384, 512, 1024])
def test_spmm_fused_relu_mask(cuda_device, width):
    """dh_spmm_csr_relu_f32: forward == SpMM+ReLU bit for bit and records the sign mask; backward with the mask ==
    SpMM(A^T, relu_backward(Y, dY)) bit for bit (same per-element op order, masked terms contribute exact zeros)."""
    from dance_amd import kernels
    a = _rand_csr(700, 700, 11, seed=width, long_row=(9, 150), empty_rows=(4, ))
    rp, c, v = _dev_csr(a, cuda_device)
    z = torch.randn(700, width, device=cuda_device)
    assert kernels.relu_mask_bytes(700, width) == 700 * (width // 128) * 16 and kernels.relu_mask_bytes(700, 50) == 0
    mask = torch.zeros(kernels.relu_mask_bytes(700, width), dtype=torch.uint8, device=cuda_device)
    y = kernels.spmm_csr_relu(rp, c, v, z, act=kernels.ACT_RELU, out_mask=mask)
    y_ref = kernels.spmm_csr(rp, c, v, z, act=kernels.ACT_RELU)
    assert torch.equal(y, y_ref)
    dy = torch.randn(700, width, device=cuda_device)
    rpt, ct, vt, _ = kernels.csr_transpose(rp, c, v, 700, 700)
    ds = kernels.spmm_csr_relu(rpt, ct, vt, dy, in_mask=mask)
    ds_ref = kernels.spmm_csr(rpt, ct, vt, kernels.relu_backward(y_ref, dy))
    assert torch.equal(ds, ds_ref)
    with pytest.raises(Exception):
        kernels.spmm_csr_relu(rp, c, v, torch.randn(700, 50, device=cuda_device), act=kernels.ACT_RELU)
    # the mask bytes are the documented layout: a plain bitmap, bit c % 32 of word c / 32 of the row = [y[row, c] > 0]
    import cpu_ops
    assert np.array_equal(cpu_ops._mask_to_bool(mask.cpu(), 700, width), (y_ref > 0).cpu().numpy())
    # rows variants: the listed rows only, everything else untouched; mask rows likewise
    rows = torch.from_numpy(np.random.default_rng(0).choice(700, 333, replace=False).astype(np.int32)).to(cuda_device)
    rest = torch.ones(700, dtype=torch.bool, device=cuda_device)
    rest[rows.long()] = False
    out = torch.full((700, width), -7.0, device=cuda_device)
    m2 = torch.zeros_like(mask)
    kernels.spmm_csr_relu(rp, c, v, z, act=kernels.ACT_RELU, out_mask=m2, out=out, rows=rows)
    assert torch.equal(out[rows.long()], y_ref[rows.long()]) and bool((out[rest] == -7.0).all())
    bpr = kernels.relu_mask_bytes(1, width)
    assert torch.equal(m2.reshape(700, bpr)[rows.long()], mask.reshape(700, bpr)[rows.long()]) and int(m2.reshape(700, bpr)[rest].sum()) == 0
    out2 = torch.full((700, width), -7.0, device=cuda_device)
    bias = torch.randn(width, device=cuda_device)
    full = kernels.spmm_csr(rp, c, v, z, bias=bias, reduce=kernels.REDUCE_MEAN)
    kernels.spmm_csr(rp, c, v, z, bias=bias, reduce=kernels.REDUCE_MEAN, out=out2, rows=rows)
    assert torch.equal(out2[rows.long()], full[rows.long()]) and bool((out2[rest] == -7.0).all())
    # packing rows for a peer, with the ReLU mask applied on the way: rows of G = dY * [Y > 0]
    g_ref = kernels.relu_backward(y_ref, dy)
    assert torch.equal(kernels.gather_rows(dy, rows, relu_mask=mask), g_ref[rows.long()])
    assert torch.equal(kernels.gather_rows(dy, rows), dy[rows.long()])
    # operand rows without a mask of their own (halo rows that arrive masked): all-ones words
    ext_mask = torch.cat((mask, torch.full((5 * bpr, ), 255, dtype=torch.uint8, device=cuda_device)))
    z_ext = torch.cat((dy, torch.randn(5, width, device=cuda_device)))
    a_ext = _rand_csr(60, 705, 9, seed=3)
    rpe, ce, ve = _dev_csr(a_ext, cuda_device)
    got = kernels.spmm_csr_relu(rpe, ce, ve, z_ext, in_mask=ext_mask)
    assert torch.equal(got, kernels.spmm_csr(rpe, ce, ve, torch.cat((g_ref, z_ext[700:]))))


def test_degenerate_sizes(cuda_device):
    """Empty inputs are legal and cheap everywhere on the ABI: no rows, no edges, no inner dimension, no queries."""
    from dance_amd import kernels
    dev = cuda_device
    i32 = lambda *v: torch.tensor(v, dtype=torch.int32, device=dev)
    # SpMM over a graph without edges / without rows
    z = torch.randn(5, 8, device=dev)
    y = kernels.spmm_csr(i32(0, 0, 0, 0), i32(), torch.empty(0, device=dev), z, bias=torch.ones(8, device=dev), act=kernels.ACT_RELU)
    assert torch.equal(y, torch.ones(3, 8, device=dev))                      # empty rows -> act(bias)
    assert kernels.spmm_csr(i32(0), i32(), None, z).shape == (0, 8)
    assert kernels.sddmm_csr(i32(0, 0), i32(), z[:1], z).numel() == 0
    # GEMM with an empty inner dimension: C = 0 (or unchanged when accumulating); empty outputs
    a, b = torch.empty(4, 0, device=dev), torch.empty(0, 6, device=dev)
    assert torch.equal(kernels.gemm(a, b), torch.zeros(4, 6, device=dev))
    acc = torch.full((4, 6), 2.0, device=dev)
    assert torch.equal(kernels.gemm(a, b, out=acc, accumulate=True), torch.full((4, 6), 2.0, device=dev))
    assert kernels.gemm(torch.empty(0, 3, device=dev), torch.randn(3, 6, device=dev)).shape == (0, 6)
    a16, b16 = torch.empty(4, 0, dtype=torch.bfloat16, device=dev), torch.empty(6, 0, dtype=torch.bfloat16, device=dev)
    out = kernels.gemm_bf16(a16, b16, trans_b=True, bias=torch.arange(6., device=dev) - 2, act=kernels.ACT_RELU, out_dtype=torch.float32)
    assert torch.equal(out, torch.relu(torch.arange(6., device=dev) - 2).expand(4, 6))
    # kNN: empty query range; k larger than the number of points pads with (-1, inf)
    x = torch.randn(6, 4, device=dev)
    idx, dist = kernels.knn(x, 3, 2, 2)
    assert idx.shape == (0, 3)
    idx, dist = kernels.knn(x, 9)
    assert bool((idx[:, 6:] == -1).all()) and bool(torch.isinf(dist[:, 6:]).all()) and bool((idx[:, 0] == torch.arange(6, device=dev)).all())
    # reductions / elementwise on empty matrices
    assert torch.equal(kernels.colsum(torch.empty(0, 5, device=dev)), torch.zeros(5, device=dev))
    assert kernels.relu_backward(torch.empty(0, 5, device=dev), torch.empty(0, 5, device=dev)).shape == (0, 5)


def test_sddmm_bf16_and_spatial_gaussian_knn(cuda_device):
    from dance_amd import kernels
    from oracle import graphs as og
    rng = np.random.default_rng(5)
    a = _rand_csr(400, 300, 9, seed=2)
    rp, c, v = _dev_csr(a, cuda_device)
    u = torch.randn(400, 72, device=cuda_device).to(torch.bfloat16)
    w = torch.randn(300, 72, device=cuda_device).to(torch.bfloat16)
    got = kernels.sddmm_csr(rp, c, u, w, scale=v)
    rows = np.repeat(np.arange(400), np.diff(a.indptr))
    ref = (u.double().cpu().numpy()[rows] * w.double().cpu().numpy()[a.indices]).sum(1) * a.data
    assert rel_err(got.cpu().numpy(), ref) < 1e-5            # bf16 inputs, fp32 products and sums
    # kNN-truncated Gaussian adjacency == kNN (oracle) + the reference's kernel expression, rows sorted by column
    xy = (rng.random((1500, 2)) * 30).astype(np.float32)
    for l in (0.0, 1.7):
        rowptr, col, val = kernels.spatial_gaussian_knn(torch.from_numpy(xy).to(cuda_device), 12, l)
        idx, dist = og.knn_exact(xy, 12)
        order = np.argsort(idx, axis=1)
        ref_col, ref_d = np.take_along_axis(idx, order, 1), np.take_along_axis(dist, order, 1)
        assert np.array_equal(rowptr.cpu().numpy(), np.arange(0, 1500 * 12 + 1, 12)) and np.array_equal(col.cpu().numpy().reshape(1500, 12), ref_col)
        want = ref_d if l <= 0 else np.exp(-(ref_d * ref_d) / np.float32(2 * l * l))
        assert rel_err(val.cpu().numpy().reshape(1500, 12), want) < 1e-6


def test_adam_step_kernel_vs_torch(cuda_device):
    """dh_adam_step_f32 (two launches for the whole parameter set) against torch.optim.Adam's single-tensor implementation on the same
    tensors, 6 steps, with and without weight decay: parameters and moments agree to fp32 rounding, the step counters are ticked."""
    from dance_amd import kernels
    for wd in (0.0, 1e-2):
        torch.manual_seed(3)
        shapes = [(37, 19), (19, ), (400, 200), (200, ), (5, 3, 2), (1, ), (1025, ), (64, 64), (3, ), (7, 11)]  # 10 tensors: two launches of <= 8
        ps = [torch.randn(s, device=cuda_device).requires_grad_(True) for s in shapes]
        qs = [p.detach().clone().requires_grad_(True) for p in ps]
        opt_k = torch.optim.Adam(ps, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd, capturable=True, foreach=False)
        opt_t = torch.optim.Adam(qs, lr=3e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=wd, capturable=True, foreach=False)
        for it in range(6):
            gs = [torch.randn_like(p) * (0.1 + it) for p in ps]
            for p, q, g in zip(ps, qs, gs):
                p.grad, q.grad = g.clone(), g.clone()
            used = kernels.adam_step(opt_k)
            assert used == (it > 0)          # the very first step creates the state: torch's own
            if not used:
                opt_k.step()
            opt_t.step()
        for p, q in zip(ps, qs):
            assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * max(1.0, float(q.detach().abs().max()))
            sk, st = opt_k.state[p], opt_t.state[q]
            assert float(sk["step"]) == float(st["step"]) == 6.0
            assert float((sk["exp_avg"] - st["exp_avg"]).abs().max()) <= 1e-6 * max(1.0, float(st["exp_avg"].abs().max()))
            assert float((sk["exp_avg_sq"] - st["exp_avg_sq"]).abs().max()) <= 4e-6 * max(1.0, float(st["exp_avg_sq"].abs().max()))  # (1 - b2) g g vs g g (1 - b2)
    # outside the kernel's coverage: nothing is touched
    p = torch.randn(4, device=cuda_device, requires_grad=True)
    p.grad = torch.ones_like(p)
    o = torch.optim.Adam([p], amsgrad=True, capturable=True)
    o.step()
    before = p.detach().clone()
    assert kernels.adam_step(o) is False and torch.equal(p, before)
    assert kernels.adam_step(torch.optim.SGD([p], lr=0.1)) is False


@pytest.mark.parametrize("M,N,K,tb", [(1000, 512, 2000, True), (257, 130, 100, False), (5000, 50, 50, True), (64, 300, 9000, True), (3, 7, 5, False),
                                      (2000, 512, 70_000, True)])
@pytest.mark.parametrize("act", [0, 1])
def test_gemm_f32_bias_act_equals_two_passes(cuda_device, M, N, K, tb, act):
    """dh_gemm_f32_bias_act (bias + ReLU in the output tile's store; split-K: in the reduce kernel; narrow shapes: behind the streaming
    kernel) is bit for bit dh_gemm_f32 followed by dh_bias_act_f32."""
    from dance_amd import kernels
    g = torch.Generator(device=cuda_device).manual_seed(M + N + K)
    a = torch.randn(M, K, device=cuda_device, generator=g)
    b = torch.randn((N, K) if tb else (K, N), device=cuda_device, generator=g) / K**0.5
    bias = torch.randn(N, device=cuda_device, generator=g)
    two = kernels.bias_act_(kernels.gemm(a, b, trans_b=tb), bias, act)
    one = kernels.gemm(a, b, trans_b=tb, bias=bias, act=act)
    assert torch.equal(one, two)
    if act:
        assert torch.equal(kernels.gemm(a, b, trans_b=tb, act=act), torch.relu(kernels.gemm(a, b, trans_b=tb)))


@pytest.mark.parametrize("n,c", [(1, 1), (1000, 16), (65537, 16), (513, 3), (300, 40), (257, 64), (130, 300), (64, 1000)])
def test_softmax_xent_sum_matches_torch(cuda_device, n, c):
    """dh_softmax_xent_sum_f32 == F.cross_entropy(reduction="sum") in float64 and its gradient, ignore_index rows included; the
    autograd wrapper scales by the upstream scalar; deterministic."""
    from dance_amd import kernels
    from dance_amd.autograd import CrossEntropySum
    g = torch.Generator(device="cpu").manual_seed(n + c)
    x = (torch.randn(n, c, generator=g) * 3).to(cuda_device).requires_grad_(True)
    y = torch.randint(0, c, (n, ), generator=g)
    if n > 10:
        y[::7] = -100
    y = y.to(cuda_device)
    loss = CrossEntropySum()(x, y)
    gx, = torch.autograd.grad(loss * 0.37, x)
    x64 = x.detach().double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(x64, y, reduction="sum")
    gref, = torch.autograd.grad(ref * 0.37, x64)
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-6
    assert rel_err(gx.cpu().numpy(), gref.cpu().numpy()) < 2e-6
    if n > 10:
        assert bool((gx[::7] == 0).all())
    l2, d2 = kernels.softmax_xent_sum(x.detach(), y)
    l3, d3 = kernels.softmax_xent_sum(x.detach(), y)
    assert torch.equal(l2, l3) and torch.equal(d2, d3)
    xs = torch.zeros(n, c + 5, device=cuda_device)   # strided rows
    xs[:, :c] = x.detach()
    l4, d4 = kernels.softmax_xent_sum(xs[:, :c], y)
    assert torch.equal(l4, l2) and torch.equal(d4, d2)


@pytest.mark.parametrize("M,N,K", [(128, 300, 200), (2128, 200, 50), (128, 200, 300), (300, 200, 128), (1, 1, 1), (33, 65, 257), (500, 16, 200), (1000, 1000, 512),
                                   (31, 97, 64), (64, 64, 129)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_f32_small_vs_float64_and_the_large_kernel(cuda_device, M, N, K, ta, tb):
    """dh_gemm_f32_small (the mini-batch steps' products: 32 x 32 tiles, whole K in one round trip) against float64 and against
    dh_gemm_f32 on the same operands, every transposition, ragged edges, strided rows, bias + ReLU in the store."""
    from dance_amd import kernels
    assert kernels.GEMM_SMALL
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn((K, M) if ta else (M, K), generator=g)
    b = torch.randn((N, K) if tb else (K, N), generator=g)
    abuf = torch.full((a.shape[0], a.shape[1] + 3), float("nan"))
    abuf[:, :a.shape[1]] = a
    ad, bd = abuf.to(cuda_device)[:, :a.shape[1]], b.to(cuda_device)
    bias = torch.randn(N, generator=g).to(cuda_device)
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    with kernels.KernelTimer() as tm, kernels.mini_batch_products():
        out = kernels.gemm(ad, bd, trans_a=ta, trans_b=tb)
        out_b = kernels.gemm(ad, bd, trans_a=ta, trans_b=tb, bias=bias, act=kernels.ACT_RELU)
        again = kernels.gemm(ad, bd, trans_a=ta, trans_b=tb)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 2e-6
    assert rel_err(out_b.cpu().numpy(), torch.relu(ref + bias.cpu().double()).numpy()) < 2e-6
    assert torch.equal(out, again)  # deterministic
    big = kernels.gemm(ad, bd, trans_a=ta, trans_b=tb)  # outside the context: the large-tile kernel, whatever the shape
    assert rel_err(out.cpu().numpy(), big.cpu().numpy()) < 2e-6
    assert len(tm.summary()) == 1
