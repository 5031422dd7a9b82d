"""GPU: BASELINE.json configurations at their FULL sizes, checked through size-independent properties and through
exact parity on sampled rows (a row of a GCN output / kNN list / aggregation depends only on that row's neighbours,
so the CPU oracle can evaluate sampled rows of a 1M-cell problem in float64 in milliseconds).

config 2: GCN 100k x 2k, k=15           headline: GCN 1M x 2k, k=15
config 3: scDeepSort cell-gene graph 1M x 2k (graph build + cell<-gene aggregation)
config 5: SpaGCN 500k spots, spatial kNN k=15, 50 -> 50 GraphConvolution
(config 4, graph-sc on 8 GPUs, shares the config-3 graph and the WeightedGraphConv kernels tested elsewhere.)
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import graphs as og
from oracle import sage as osg

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand_graph(n, k, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    col = torch.randint(0, n, (n, k), device=DEV, generator=g).sort(dim=1).values.to(torch.int32).reshape(-1)
    rowptr = torch.arange(0, n * k + 1, k, dtype=torch.int32, device=DEV)
    val = torch.rand(n * k, device=DEV, generator=g) / k
    return rowptr, col, val


@pytest.mark.parametrize("n", [100_000, 1_000_000])
def test_gcn_layer_full_size(cuda_device, n):
    from dance_amd import kernels
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.single_modality.clustering.scdsc import GNNLayer
    fin, fout, k = 2000, 512, 15
    gen = torch.Generator(device=DEV).manual_seed(n)
    x = torch.randn(n, fin, device=DEV, generator=gen)
    rowptr, col, val = _rand_graph(n, k, 1)
    graph = CSRGraph(rowptr, col, val, n, n)
    layer = GNNLayer(fin, fout).to(DEV)
    dy = torch.randn(n, fout, device=DEV, generator=gen)
    y = layer(x, graph)
    y.backward(dy)
    # (1) exact parity on sampled rows, float64 oracle arithmetic: y_i = relu(sum_e a_e (x_src W))
    rows = np.random.default_rng(0).choice(n, 64, replace=False)
    w64 = layer.weight.detach().double().cpu().numpy()
    rp, c, v = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    ref = np.stack([np.maximum(sum(float(v[e]) * (x[int(c[e])].double().cpu().numpy() @ w64) for e in range(rp[i], rp[i + 1])), 0)
                    for i in rows])
    assert rel_err(y[torch.from_numpy(rows).to(DEV)].detach().cpu().numpy(), ref) < 1e-4
    # (2) transpose round trip is the identity, bit for bit; A^T has the same nnz and column sums become row counts
    gt = graph.transpose()
    rp2, c2, v2, _ = kernels.csr_transpose(gt.rowptr, gt.col, gt.val, n, n)
    assert torch.equal(rp2, rowptr) and torch.equal(c2, col) and torch.equal(v2, val)
    # (3) checksum of checksums for dW = X^T dS with dS = A^T (dy * [y > 0]):  1^T dW 1 = (X 1)^T (dS 1)
    g = torch.where(y.detach() > 0, dy, torch.zeros_like(dy))
    ds = kernels.spmm_csr(gt.rowptr, gt.col, gt.val, g)
    lhs = float(layer.weight.grad.double().sum())
    rhs = float((x.double().sum(1) * ds.double().sum(1)).sum())
    assert abs(lhs - rhs) < 1e-6 * float((x.double().abs().sum(1) * ds.double().abs().sum(1)).sum())
    # (3b) dW element by element on sampled genes x hidden columns, float64 on the device: the 64-slice split-K TN GEMM
    # at K = n is otherwise only covered at K <= 100 001 (test_gpu_kernels)
    genes = torch.from_numpy(np.random.default_rng(1).choice(fin, 24, replace=False)).to(DEV)
    cols = torch.from_numpy(np.random.default_rng(2).choice(fout, 16, replace=False)).to(DEV)
    ref_dw = x[:, genes].double().t() @ ds[:, cols].double()
    got_dw = layer.weight.grad[genes][:, cols]
    assert rel_err(got_dw.cpu().numpy(), ref_dw.cpu().numpy()) < 1e-4
    assert float((got_dw.double() - ref_dw).abs().max() / layer.weight.grad.abs().max()) < 1e-5
    # (4) dS itself: sampled rows of A^T g against the explicit edge list
    tr, tc, tv = gt.rowptr.cpu().numpy(), gt.col.cpu().numpy(), gt.val.cpu().numpy()
    ref_ds = np.stack([sum((float(tv[e]) * g[int(tc[e])].double().cpu().numpy() for e in range(tr[i], tr[i + 1])), np.zeros(fout)) for i in rows])
    assert rel_err(ds[torch.from_numpy(rows).to(DEV)].cpu().numpy(), ref_ds) < 1e-4
    # (5) linearity of the un-activated layer: f(2x) = 2 f(x) exactly in fp32 (power-of-two scaling)
    with torch.no_grad():
        a = layer(x[:50_000].contiguous(), CSRGraph(*_rand_graph(50_000, k, 2), 50_000, 50_000), active=False)
        b = layer((2 * x[:50_000]).contiguous(), CSRGraph(*_rand_graph(50_000, k, 2), 50_000, 50_000), active=False)
    assert torch.equal(b, 2 * a)


def test_scdeepsort_graph_full_size(cuda_device):
    """config 3: 1M cells x 2k genes at 10 % density (nnz = 2e8): graph build + cell<-gene aggregation."""
    from dance_amd import kernels
    n_cells, n_genes, per, d = 1_000_000, 2000, 200, 400
    gen = torch.Generator(device=DEV).manual_seed(3)
    col = torch.rand(n_cells, n_genes, device=DEV, generator=gen).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32).reshape(-1)
    rp_x = torch.arange(0, n_cells * per + 1, per, dtype=torch.int32, device=DEV)
    val_x = torch.rand(n_cells * per, device=DEV, generator=gen) + 0.5
    nnz = n_cells * per
    rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col, val_x, n_cells, n_genes)
    vx, vt = kernels.csr_row_normalize(rp_x, val_x), kernels.csr_row_normalize(rp_t, val_t)
    rowptr, gcol, gval, eid = kernels.cellgene_graph_assemble(rp_x, col, vx, rp_t, col_t, vt, perm_t, n_cells, n_genes)
    n_nodes, n_edges = n_cells + n_genes, 2 * nnz + n_cells + n_genes
    assert int(rowptr[-1]) == n_edges == gcol.numel()
    # eid is a permutation of the reference edge ids
    assert torch.equal(torch.sort(eid.to(torch.int64)).values, torch.arange(n_edges, device=DEV))
    # normalisation property (cell_feature_graph.py:62-68): in-edge weights of a node sum to its in-degree; + 1 self loop
    deg = (rowptr[1:] - rowptr[:-1]).double()
    sums = torch.zeros(n_nodes, dtype=torch.float64, device=DEV).index_add_(
        0, torch.repeat_interleave(torch.arange(n_nodes, device=DEV), (rowptr[1:] - rowptr[:-1]).long()), gval.double())
    assert float((sums - deg).abs().max() / deg.max()) < 1e-5
    # every cell row: its genes (ascending) then itself
    r = 123_456
    s, t = int(rowptr[n_genes + r]), int(rowptr[n_genes + r + 1])
    assert torch.equal(gcol[s:t - 1], col[r * per:(r + 1) * per]) and int(gcol[t - 1]) == n_genes + r
    # cell<-gene aggregation on all 1M cell rows; sampled rows vs the oracle restatement of gnn.py:62-82,90
    feats = torch.randn(n_nodes, d, device=DEV, generator=gen)
    cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(DEV)
    alpha = torch.rand(n_genes + 2, device=DEV, generator=gen) + 0.5
    neigh = kernels.sage_aggregate(rowptr[n_genes:], gcol, gval, cid, cid[n_genes:], alpha, feats)
    rows = np.random.default_rng(1).choice(n_cells, 32, replace=False)
    a_np, cid_np = alpha.cpu().numpy(), cid.cpu().numpy()
    for i in rows:
        s, t = int(rowptr[n_genes + i]), int(rowptr[n_genes + i + 1])
        src = gcol[s:t].cpu().numpy().astype(np.int64)
        ref = osg.sage_neigh(np.arange(t - s), np.zeros(t - s, dtype=np.int64), gval[s:t].cpu().numpy(), cid_np[src], cid_np[[n_genes + i]],
                             a_np, feats[torch.from_numpy(src).to(DEV)].cpu().numpy(), 1)
        assert rel_err(neigh[i].cpu().numpy(), ref[0]) < 1e-5
    # config 3 proper: bf16 storage.  The bf16 kernel with fp32 output equals the fp32 kernel on the rounded features
    # bit for bit (same edge order, same fmaf chain); the dense update runs on the bf16 matrix cores.
    feats16 = feats.to(torch.bfloat16)
    n32 = kernels.sage_aggregate(rowptr[n_genes:], gcol, gval, cid, cid[n_genes:], alpha, feats16.float())
    n16 = kernels.sage_aggregate_bf16(rowptr[n_genes:], gcol, gval, cid, cid[n_genes:], alpha, feats16, out_dtype=torch.float32)
    assert torch.equal(n16, n32)
    del n32, neigh
    w16 = (torch.randn(200, d, device=DEV, generator=gen) / 20).to(torch.bfloat16)
    bias = torch.randn(200, device=DEV, generator=gen)
    h16 = feats16[n_genes:]
    z = kernels.gemm_bf16(h16, w16, trans_b=True, bias=bias, act=kernels.ACT_RELU)
    assert z.shape == (n_cells, 200) and z.dtype == torch.bfloat16
    sel = torch.from_numpy(np.sort(rows)).to(DEV)
    ref = torch.relu(h16[sel].double() @ w16.double().T + bias.double())
    assert rel_err(z[sel].float().cpu().numpy(), ref.cpu().numpy()) < 2**-8  # one bf16 rounding of an fp32 accumulation


def test_knn_filter_full_size(cuda_device):
    """NeighborGraph's kNN at the headline size (1M cells x 50 PCs, k = 15) through the matrix-core filter; sampled query
    ranges against the exact vector-ALU scan (bit for bit), plus size-independent properties of the whole result."""
    from dance_amd import kernels
    n, d, k = 1_000_000, 50, 15
    gen = torch.Generator(device=DEV).manual_seed(11)
    centers = torch.randn(64, d, device=DEV, generator=gen) * 3
    x = centers[torch.randint(0, 64, (n, ), device=DEV, generator=gen)] + torch.randn(n, d, device=DEV, generator=gen)
    idx, dist = kernels.knn(x, k)  # auto -> filter
    assert torch.equal(idx[:, 0], torch.arange(n, device=DEV, dtype=torch.int32)) and float(dist[:, 0].abs().max()) == 0.0  # self first
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())                                                  # sorted by distance
    assert int(idx.min()) >= 0 and int(idx.max()) < n
    assert bool((torch.sort(idx, dim=1).values[:, 1:] != torch.sort(idx, dim=1).values[:, :-1]).all())   # no duplicates
    for q0 in (0, 499_744, n - 256):
        i_s, d_s = kernels.knn(x, k, q0, q0 + 256, algo=kernels.KNN_SCAN)
        assert torch.equal(idx[q0:q0 + 256], i_s) and torch.equal(dist[q0:q0 + 256], d_s)


def test_spagcn_full_size(cuda_device):
    """config 5: 500k spots on a jittered hex grid, spatial kNN (k=15) truncated Gaussian adjacency, 50 -> 50 layer."""
    from dance_amd import kernels
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.spatial.spatial_domain.spagcn import GraphConvolution, SpaGCN
    n, k = 500_000, 15
    rng = np.random.default_rng(5)
    side = int(np.ceil(np.sqrt(n)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = np.stack([gx.ravel() + 0.5 * (gy.ravel() % 2), gy.ravel() * 0.866], 1)[:n] + rng.normal(0, 0.05, (n, 2))
    xyz = np.hstack([xy, rng.normal(0, 0.3, (n, 1))]).astype(np.float32)
    x_dev = torch.from_numpy(xyz).to(DEV)
    idx, dist = kernels.knn(x_dev, k)
    # exact parity of sampled queries against brute force over all 500k points (bit-defined distance, ties -> lower index)
    q = rng.choice(n, 48, replace=False)
    acc = np.zeros((q.size, n), dtype=np.float32)
    for t in range(3):
        diff = xyz[q, t, None] - xyz[None, :, t]
        acc = acc + diff * diff
    order = np.argsort(acc, axis=1, kind="stable")[:, :k]
    assert np.array_equal(idx[torch.from_numpy(q).to(DEV)].cpu().numpy(), order)
    assert np.array_equal(dist[torch.from_numpy(q).to(DEV)].cpu().numpy(), np.sqrt(np.take_along_axis(acc, order, 1)))
    # sortedness / self-first properties on every row
    assert bool((dist[:, 1:] >= dist[:, :-1]).all()) and torch.equal(idx[:, 0], torch.arange(n, dtype=torch.int32, device=DEV))
    order_t = torch.argsort(idx, dim=1)
    g = CSRGraph(torch.arange(0, n * k + 1, k, dtype=torch.int32, device=DEV), torch.gather(idx, 1, order_t).reshape(-1).contiguous(),
                 torch.gather(dist, 1, order_t).reshape(-1).contiguous(), n, n)
    model = SpaGCN(l=1.2, device=DEV)
    adj_exp = model.calc_adj_exp(g)
    assert rel_err(adj_exp.val[:1000].cpu().numpy(), np.exp(-(g.val[:1000].cpu().numpy().astype(np.float64)**2) / (2 * 1.2**2))) < 1e-5
    layer = GraphConvolution(50, 50).to(DEV)
    emb = torch.randn(n, 50, device=DEV)
    out = layer(emb, adj_exp)
    out.sum().backward()
    rows = rng.choice(n, 64, replace=False)
    w64, b64 = layer.weight.detach().double().cpu().numpy(), layer.bias.detach().double().cpu().numpy()
    rp, c, v = adj_exp.rowptr.cpu().numpy(), adj_exp.col.cpu().numpy(), adj_exp.val.cpu().numpy()
    ref = np.stack([sum(float(v[e]) * (emb[int(c[e])].double().cpu().numpy() @ w64) for e in range(rp[i], rp[i + 1])) + b64 for i in rows])
    assert rel_err(out[torch.from_numpy(rows).to(DEV)].detach().cpu().numpy(), ref) < 1e-4
    # bias gradient of sum(out) is exactly the number of spots
    assert rel_err(layer.bias.grad.cpu().numpy(), np.full(50, float(n))) < 1e-6


def test_sage_mfma_matches_gather_at_scale(cuda_device):
    """dh_sage_window_mfma against the gather kernel on a CellFeatureGraph-shaped graph of 200k cells x 2000 genes (10 % dense,
    D = 400): every row, 1e-5 of the max-norm; bf16 features: the fp32-output results agree to the same bar."""
    from dance_amd import kernels
    dev, n_cells, n_genes, dfeat, per = cuda_device, 200_000, 2000, 400, 200
    g = torch.Generator(device=dev).manual_seed(0)
    col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32)
    col = torch.cat((col, (n_genes + torch.arange(n_cells, device=dev, dtype=torch.int32))[:, None]), 1).reshape(-1).contiguous()
    rowptr = torch.arange(0, n_cells * (per + 1) + 1, per + 1, dtype=torch.int32, device=dev)
    w = torch.rand(col.numel(), device=dev, generator=g) + 0.5
    feats = torch.randn(n_genes + n_cells, dfeat, device=dev, generator=g)
    cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
    alpha = torch.rand(n_genes + 2, device=dev, generator=g) + 0.5
    args = (rowptr, col, w, cid, cid[n_genes:].contiguous(), alpha)
    ref = kernels.sage_aggregate(*args, feats)
    got = kernels.sage_aggregate_mfma(*args, feats, 0, n_genes)
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
    f16 = feats.to(torch.bfloat16)
    ref16 = kernels.sage_aggregate(*args, f16.float())
    got16 = kernels.sage_aggregate_mfma(*args, f16, 0, n_genes, out_dtype=torch.float32)
    assert float((got16 - ref16).abs().max() / ref16.abs().max()) < 1e-5


def _cellgene_graph(n_cells, n_genes, per, d, seed):
    from dance_amd import kernels
    from dance_amd.cellgraph import CellGeneGraph
    g = torch.Generator(device=DEV).manual_seed(seed)
    col = torch.rand(n_cells, n_genes, device=DEV, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32).reshape(-1)
    rp_x = torch.arange(0, n_cells * per + 1, per, dtype=torch.int32, device=DEV)
    val_x = torch.rand(n_cells * per, device=DEV, generator=g) + 0.5
    rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col, val_x, n_cells, n_genes)
    rowptr, gcol, gval, eid = kernels.cellgene_graph_assemble(rp_x, col, val_x, rp_t, col_t, val_t, perm_t, n_cells, n_genes)
    n_nodes = n_cells + n_genes
    cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(DEV)
    fid = torch.cat((-torch.ones(n_genes, dtype=torch.int32), torch.arange(n_cells, dtype=torch.int32))).to(DEV)
    feats = torch.randn(n_nodes, d, device=DEV, generator=g)
    return CellGeneGraph(rowptr, gcol, gval, eid, n_nodes, {"cell_id": cid, "feat_id": fid, "features": feats})


@pytest.mark.parametrize("batch", [128, 8192])
def test_graphsc_batch_full_size(cuda_device, batch):
    """config 4: graph-sc on the 1M-cell x 2k-gene cell-gene graph (nnz 2e8, D = 50 -> 200 -> 300).  One mini-batch at the
    reference's batch size (128, graphsc.py:155) and at the throughput batch (8192): the block of the full-neighbour sampler,
    WeightedGraphConv on it (sampled destination rows against the float64 restatement of graphsc.py:428-484: block-local
    D_out^-1/2, W before the weighted sum, D_in^-1/2, bias, ReLU), and the decoder loss of GraphSC.fit (fused, no B x B logits)
    against the float64 BCE-with-logits of graphsc.py:208-216 on the same embedding."""
    import torch.nn.functional as F
    from dance_amd.cellgraph import DataLoader, MultiLayerFullNeighborSampler
    from dance_amd.modules.single_modality.clustering import graphsc
    n_cells, n_genes, per, d = 1_000_000, 2000, 200, 50
    cg = _cellgene_graph(n_cells, n_genes, per, d, seed=4)
    cg.ndata["order"] = cg.ndata["label"] = cg.ndata["feat_id"]
    torch.manual_seed(0)
    model = graphsc.GCNAE(agg="sum", activation="relu", in_feats=d, n_hidden=1, hidden_dim=200, hidden_1=300, hidden_2=0, dropout=0.0,
                          n_layers=1, hidden_relu=False, hidden_bn=False).to(DEV)
    seeds = np.arange(n_genes, n_genes + n_cells)
    loader = DataLoader(cg, seeds, MultiLayerFullNeighborSampler(1), batch_size=batch, shuffle=True, drop_last=False,
                        generator=torch.Generator().manual_seed(9), block_hook=graphsc._dst_edge_hook)
    input_nodes, output_nodes, blocks = next(iter(loader))
    blk = blocks[0]
    b = blk.number_of_dst_nodes()
    assert b == batch and blk.number_of_src_nodes() <= n_genes + batch and blk.number_of_edges() == batch * (per + 1)
    x = blk.srcdata["features"]
    h = model.layer1(blk, x, agg="sum")
    # float64 restatement on sampled destination rows, from the block's own CSR
    rp, col, val = blk.rowptr.cpu().numpy(), blk.col.cpu().numpy().astype(np.int64), blk.val.cpu().numpy().astype(np.float64)
    out_deg = np.maximum(np.bincount(col, minlength=blk.number_of_src_nodes()), 1).astype(np.float64)
    w64, b64 = model.layer1.weight.detach().double().cpu().numpy(), model.layer1.bias.detach().double().cpu().numpy()
    x64 = x.double().cpu().numpy()
    rows = np.random.default_rng(0).choice(b, 48, replace=False)
    ref = []
    for i in rows:
        e = slice(rp[i], rp[i + 1])
        src = col[e]
        msg = ((x64[src] * out_deg[src, None]**-0.5) @ w64) * val[e, None]
        ref.append(np.maximum(msg.sum(0) * max(rp[i + 1] - rp[i], 1)**-0.5 + b64, 0))
    assert rel_err(h[torch.from_numpy(rows).to(DEV)].detach().cpu().numpy(), np.stack(ref)) < 1e-4
    # decoder loss: fused path of GraphSC.fit vs float64 dense BCE-with-logits with pos_weight / norm of graphsc.py:208-216
    _, emb = model.forward(blocks, x, decode=False)
    order_e, cnt, ev = blk.hook_out
    if ev is not None:
        ev.synchronize()
    n_listed = int(cnt)
    sel = order_e[:n_listed]
    us = blk.col[sel].to(torch.int64)
    vs = torch.searchsorted(blk.rowptr, sel.to(blk.rowptr.dtype), right=True).to(torch.int64) - 1
    assert n_listed == b  # cell-gene graph: the only edges among the batch's cells are their self loops
    total = float(b)**2
    pos_weight, norm = (total - n_listed) / n_listed, total / ((total - n_listed) * 2)
    emb = emb.detach().requires_grad_(True)
    loss = norm * graphsc.gram_listed_bce(emb, us, vs, pos_weight)
    loss.backward()
    z64 = emb.detach().double().requires_grad_(True)
    adj = torch.zeros((b, b), dtype=torch.float64, device=DEV)
    adj[us, vs] = 1.0
    ref_loss = norm * F.binary_cross_entropy_with_logits(z64 @ z64.T, adj, pos_weight=torch.tensor(pos_weight, dtype=torch.float64, device=DEV))
    ref_loss.backward()
    assert abs(float(loss) - float(ref_loss)) < 1e-5 * abs(float(ref_loss))
    assert rel_err(emb.grad.cpu().numpy(), z64.grad.cpu().numpy()) < 1e-4


def test_sage_all_nodes_full_size(cuda_device):
    """Config 3's graph at 1M cells x 2000 genes: the aggregation of EVERY node — gene rows through dh_sage_window_splitk (2000 rows of
    ~1e5 cell in-edges), cell rows through dh_sage_window_mfma_planned — against the gather kernel on all rows (max-norm 2e-5), and
    two size-independent properties of the gene rows: linearity in the features and a row-weighted checksum (sum_v deg(v) neigh[v] =
    sum over edges of alpha w h[src], evaluated independently as one SpMM of the transposed graph)."""
    from dance_amd import kernels
    cg = _cellgene_graph(1_000_000, 2000, 200, 400, 5)
    n_genes = cg.gene_prefix()
    n_cells = cg.number_of_nodes() - n_genes
    cid, h = cg.ndata["cell_id"], cg.ndata["features"]
    alpha = torch.rand(n_genes + 2, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) + 0.5
    args = (cg.rowptr, cg.col, cg.val, cid, cid, alpha)
    ref = kernels.sage_aggregate(*args, h)
    rp_g = cg.rowptr[:n_genes + 1]
    genes = kernels.sage_aggregate_splitk(rp_g, cg.col, cg.val, cid, cid[:n_genes], alpha, h, n_genes, n_cells)
    cells = kernels.sage_aggregate_mfma(cg.rowptr[n_genes:], cg.col, cg.val, cid, cid[n_genes:].contiguous(), alpha, h, 0, n_genes)
    scale = float(ref.abs().max())
    assert float((genes - ref[:n_genes]).abs().max()) < 2e-5 * scale
    assert float((cells - ref[n_genes:]).abs().max()) < 2e-5 * scale
    # linearity: aggregate(2 h + h') = 2 aggregate(h) + aggregate(h')
    h2 = torch.randn(h.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    lin = kernels.sage_aggregate_splitk(rp_g, cg.col, cg.val, cid, cid[:n_genes], alpha, 2 * h + h2, n_genes, n_cells)
    g2 = kernels.sage_aggregate_splitk(rp_g, cg.col, cg.val, cid, cid[:n_genes], alpha, h2, n_genes, n_cells)
    assert float((lin - (2 * genes + g2)).abs().max()) < 4e-5 * float(lin.abs().max())
    # checksum of the gene rows: sum_v deg(v) neigh[v] == (edge values alpha[.] w as a row vector) . H through the gene rows' own CSR
    deg = (rp_g[1:] - rp_g[:-1]).double()
    lhs = (deg[:, None] * genes.double()).sum(0)
    rows = torch.repeat_interleave(torch.arange(n_genes, device=DEV), (rp_g[1:] - rp_g[:-1]).long())
    e_hi = int(rp_g[-1])
    src = cg.col[:e_hi].long()
    sid, did = cid[src].long(), cid[:n_genes][rows].long()
    idx = torch.where(sid >= 0, torch.full_like(sid, n_genes), did)   # gene destinations: cell source -> alpha[gene id], gene source -> alpha[G]
    coef = (alpha.double()[idx] * cg.val[:e_hi].double())
    colsum = torch.zeros(cg.number_of_nodes(), dtype=torch.float64, device=DEV).index_add_(0, src, coef)
    rhs = colsum @ h.double()
    assert float((lhs - rhs).abs().max()) < 2e-5 * float(rhs.abs().max())


def test_zinb_full_size_properties(cuda_device):
    """dh_zinb_nll_* at 1M x 2000 (config 1's scDSC / scTAG loss at the headline's cell count): the loss of the whole matrix equals the
    row-count-weighted mean of the losses of its two halves (the kernel's float64 row sums are additive), every gradient entry is
    finite, and sampled rows agree with the float64 formula."""
    import cpu_ops
    from dance_amd import autograd
    n, gz = 1_000_000, 2000
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.poisson(torch.rand(n, gz, device=DEV, generator=g) * 2.0) * (torch.rand(n, gz, device=DEV, generator=g) < 0.2)
    mean = (torch.rand(n, gz, device=DEV, generator=g) * 4 + 1e-3).requires_grad_(True)
    disp = (torch.rand(n, gz, device=DEV, generator=g) * 3 + 1e-3).requires_grad_(True)
    pi = (torch.rand(n, gz, device=DEV, generator=g) * 0.98 + 0.01).requires_grad_(True)
    sf = torch.rand(n, device=DEV, dtype=torch.float64, generator=g) + 0.5
    whole = autograd.zinb_nll(x, mean, disp, pi, sf)
    gm, gd, gp = torch.autograd.grad(whole, (mean, disp, pi))
    cut = 333_337
    a = autograd.zinb_nll(x[:cut], mean[:cut].detach(), disp[:cut].detach(), pi[:cut].detach(), sf[:cut])
    b = autograd.zinb_nll(x[cut:], mean[cut:].detach(), disp[cut:].detach(), pi[cut:].detach(), sf[cut:])
    assert abs(float(whole) - (float(a) * cut + float(b) * (n - cut)) / n) < 1e-12 * abs(float(whole))
    for t in (gm, gd, gp):
        assert bool(torch.isfinite(t).all())
    rows = torch.randint(0, n, (4000, ), device=DEV, generator=g)
    m64, d64, p64 = (t[rows].detach().double().requires_grad_(True) for t in (mean, disp, pi))
    ref = cpu_ops._zinb_elements(x[rows], m64, d64, p64, sf[rows], 0.0).sum() / (n * gz)   # these rows' share of the mean over all elements
    rm, rd, rp = torch.autograd.grad(ref, (m64, d64, p64))
    for got, want in ((gm[rows], rm), (gd[rows], rd), (gp[rows], rp)):
        assert float((got.double() - want).abs().max()) < 1e-6 * float(want.abs().max())


def test_bf16_weight_gradient_full_size(cuda_device):
    """C3's dW = dY^T X over 1M cells (gemm_bf16_tn_tall_kernel): additivity over row ranges — the product over all rows equals the
    sum of the products over two ragged parts (each through the same kernel) to fp32 rounding — and equality with the float64
    product on a 20k-row slice."""
    from dance_amd import kernels
    g = torch.Generator(device=DEV).manual_seed(3)
    k = 1_000_003
    dy = torch.randn(k, 200, device=DEV, generator=g).to(torch.bfloat16)
    x = torch.randn(k, 400, device=DEV, generator=g).to(torch.bfloat16)
    whole = kernels.gemm_bf16(dy, x, trans_a=True, out_dtype=torch.float32)
    cut = 400_001
    parts = kernels.gemm_bf16(dy[:cut], x[:cut], trans_a=True, out_dtype=torch.float32) + kernels.gemm_bf16(dy[cut:], x[cut:], trans_a=True,
                                                                                                            out_dtype=torch.float32)
    assert float((whole - parts).abs().max()) < 2e-6 * float(whole.abs().max()) * 30  # ~sqrt(K) growth of fp32 partial sums
    small = kernels.gemm_bf16(dy[:20_000], x[:20_000], trans_a=True, out_dtype=torch.float32)
    ref = dy[:20_000].double().t() @ x[:20_000].double()
    assert float((small.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())


def test_captured_fits_survive_the_eager_last_batch(cuda_device, monkeypatch):
    """(The hipGraph path — since round 6 the fallback for shapes the persistent step does not cover; forced here.)  Regression (round 4): three epochs of GraphSC.fit at batch 128 and of ScDeepSort.fit at batch 500 on 100k cells, every full
    batch replayed from a captured hipGraph and the short last batch of each epoch run eagerly.  The eager batch's
    ``optimizer.zero_grad()`` used to drop the last reference to the gradient tensors the graph writes; at this size the allocator
    then released their memory and the next epoch's replays died with a GPU memory access fault (toy sizes never released the
    block).  The captured step now holds those tensors (``capture.CapturedStep.kept``)."""
    import tempfile

    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    from dance_amd.modules.single_modality.cell_type_annotation import scdeepsort as sds_mod
    from dance_amd.modules.single_modality.clustering import graphsc as gsc_mod
    monkeypatch.setattr(gsc_mod, "MINISTEP", False)
    monkeypatch.setattr(sds_mod, "MINISTEP", False)
    n_cells = 100_000
    cg = _cellgene_graph(n_cells, 2000, 200, 50, seed=6)
    m = GraphSC(in_feats=50, n_clusters=10, device="cuda")
    m.fit(cg, epochs=3, batch_size=128)                                   # 781 captured steps + a batch of 32 per epoch
    assert m.step_mode == "hipgraph"
    assert len(m.losses) == 3 * 782 and np.isfinite(m.losses).all() and m.get_latent().shape[0] == n_cells
    labels = torch.randint(0, 8, (n_cells, ), generator=torch.Generator().manual_seed(0))
    with tempfile.TemporaryDirectory() as tmp:
        s = ScDeepSort(50, 32, 1, "synthetic", "reg", batch_size=500, device="cuda", save_root=tmp, verbose=False)
        s.fit(cg, labels, epochs=3, lr=1e-3, val_ratio=0.2)                # 80 000 training cells: 160 captured steps, no tail ...
        assert s._captured is not None and len(s._captured[0].kept) > 0
        s2 = ScDeepSort(50, 32, 1, "synthetic", "reg2", batch_size=500, device="cuda", save_root=tmp, verbose=False)
        s2.fit(cg, labels, epochs=3, lr=1e-3, val_ratio=0.2037)            # ... and with a short last batch
        assert np.isfinite(s2.predict_proba(cg)).all()


def test_ministep_fits_at_the_reference_batch_sizes(cuda_device):
    """The default path since round 6: three epochs of GraphSC.fit at batch 128 (781 persistent steps + an eager batch of 32 per epoch) and
    of ScDeepSort.fit at batch 500 on 100k cells x 2000 genes, the eager last batch sharing the optimiser state with the fused update."""
    import tempfile

    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    n_cells = 100_000
    cg = _cellgene_graph(n_cells, 2000, 200, 50, seed=6)
    m = GraphSC(in_feats=50, n_clusters=10, device="cuda")
    m.fit(cg, epochs=3, batch_size=128)
    assert m.step_mode == "ministep"
    assert len(m.losses) == 3 * 782 and np.isfinite(m.losses).all() and m.get_latent().shape[0] == n_cells and np.isfinite(m.get_latent()).all()
    assert np.mean(m.losses[-100:]) < np.mean(m.losses[:100])   # it trains
    labels = torch.randint(0, 8, (n_cells, ), generator=torch.Generator().manual_seed(0))
    with tempfile.TemporaryDirectory() as tmp:
        for ratio in (0.2, 0.2037):  # without and with a short last batch
            s = ScDeepSort(50, 32, 1, "synthetic", "mini", batch_size=500, device="cuda", save_root=tmp, verbose=False)
            s.fit(cg, labels, epochs=3, lr=1e-3, val_ratio=ratio)
            assert s._use_mini and s._stepper is not None and np.isfinite(s.predict_proba(cg)).all()


def test_splitk_window_wider_than_one_round_of_sets(cuda_device):
    """A cell window of 2.5M columns at D = 400 fp32: the chunk-pointer table of 1/16 of the slices no longer fits LDS next to the
    images, so the kernel takes more sets than CUs / row blocks (several rounds of workgroups).  Sparse rows (0.2 %) keep it small;
    checked against the gather kernel."""
    from dance_amd import kernels
    n_genes, n_cells, d = 2000, 2_500_000, 400
    g = torch.Generator(device=DEV).manual_seed(0)
    per = 5000
    cols = torch.randint(0, n_cells, (n_genes, per), device=DEV, generator=g).sort(dim=1).values
    keep = torch.ones_like(cols, dtype=torch.bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]                       # distinct columns per row
    counts = keep.sum(1)
    rowptr = torch.zeros(n_genes + 1, dtype=torch.int64, device=DEV)
    rowptr[1:] = torch.cumsum(counts + 1, 0)
    nnz = int(rowptr[-1])
    col = torch.empty(nnz, dtype=torch.int32, device=DEV)
    is_self = torch.zeros(nnz, dtype=torch.bool, device=DEV)
    is_self[rowptr[:-1]] = True
    col[is_self] = torch.arange(n_genes, dtype=torch.int32, device=DEV)
    col[~is_self] = (cols[keep] + n_genes).to(torch.int32)
    w = torch.rand(nnz, device=DEV, generator=g) + 0.25
    rowptr = rowptr.to(torch.int32)
    cid = torch.cat((torch.randperm(n_genes, device=DEV, generator=g).to(torch.int32), -torch.ones(n_cells, dtype=torch.int32, device=DEV)))
    alpha = torch.rand(n_genes + 2, device=DEV, generator=g) + 0.5
    h = torch.randn(n_genes + n_cells, d, device=DEV, generator=g)
    args = (rowptr, col, w, cid, cid[:n_genes].contiguous(), alpha, h)
    assert kernels.sage_splitk_supported(n_genes, n_cells, d, h.dtype, nnz)
    got = kernels.sage_aggregate_splitk(*args, n_genes, n_cells)
    ref = kernels.sage_aggregate(*args)
    assert float((got - ref).abs().max()) < 2e-5 * float(ref.abs().max())
