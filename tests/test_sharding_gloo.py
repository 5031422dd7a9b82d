"""CPU, world_size 2 (gloo): the destination-range sharded GCN layer reproduces the single-process oracle.
Compute is injected from tests/cpu_ops.py (oracle-backed); what is under test is the partitioning, the
all-gather row placement, the transposed shard and the gradient all-reduce of dance_amd/sharding.py."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_err
from oracle import layers as ol


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(n, fin, fout, k, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, fin)).astype(np.float32)
    w = (rng.standard_normal((fin, fout)) / np.sqrt(fin)).astype(np.float32)
    b = rng.standard_normal(fout).astype(np.float32)
    dy = rng.standard_normal((n, fout)).astype(np.float32)
    adj = sp.random(n, n, density=k / n, random_state=seed, format="csr", dtype=np.float32)
    adj.data = rng.uniform(0.1, 1, adj.nnz).astype(np.float32)
    adj.sort_indices()
    return x, w, b, dy, adj


def _worker(rank, world, port, n, fin, fout, k, seed, use_bias, active, q, mode="allgather", halo_dtype="f32"):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    from dance_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, w, b, dy, adj = _problem(n, fin, fout, k, seed)
        at = adj.T.tocsr()
        at.sort_indices()
        ranges, _ = sharding.row_ranges(n, world)
        lo, hi = ranges[rank]
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
        a_sh = sharding.slice_rows(t(adj.indptr, np.int32), t(adj.indices, np.int32), t(adj.data, np.float32), lo, hi, n)
        at_sh = sharding.slice_rows(t(at.indptr, np.int32), t(at.indices, np.int32), t(at.data, np.float32), lo, hi, n)
        full = None
        if mode == "alltoall":
            full = (sharding.slice_rows(t(adj.indptr, np.int32), t(adj.indices, np.int32), t(adj.data, np.float32), 0, n, n),
                    sharding.slice_rows(t(at.indptr, np.int32), t(at.indices, np.int32), t(at.data, np.float32), 0, n, n))
        sg = sharding.ShardedGCNGraph(a_sh, at_sh, n, mode=mode, full=full, halo_dtype=halo_dtype)
        xl = torch.from_numpy(x[lo:hi].copy()).requires_grad_(True)
        wt = torch.from_numpy(w.copy()).requires_grad_(True)
        bt = torch.from_numpy(b.copy()).requires_grad_(True) if use_bias else None
        y = sharding.sharded_gcn_layer(xl, wt, sg, bt, active, ops=cpu_ops)
        y.backward(torch.from_numpy(dy[lo:hi].copy()))
        extra = None
        if mode == "halo":  # what travelled: exactly the distinct remote columns of this rank's rows of A and of A^T
            need = lambda m: np.unique(m[lo:hi].indices[(m[lo:hi].indices < lo) | (m[lo:hi].indices >= hi)])
            assert np.array_equal(sg.halo.remote_ids.numpy(), need(adj)) and np.array_equal(sg.halo_t.remote_ids.numpy(), need(at))
            assert sorted(sg.halo.interior.tolist() + sg.halo.boundary.tolist()) == list(range(hi - lo))
            assert sg.stats["exchanges"] == (2 if world > 1 else 0)
            # the single-process (emulated) plan of this rank = the plan the collective produced
            from dance_amd.graph import CSRGraph
            g_all = CSRGraph(t(adj.indptr, np.int32), t(adj.indices, np.int32), t(adj.data, np.float32), n, n)
            g_all._t = CSRGraph(t(at.indptr, np.int32), t(at.indices, np.int32), t(at.data, np.float32), n, n)
            emu = sharding.ShardedGCNGraph.from_global_csr(g_all, mode="halo", emulate=(rank, world))
            for real, fake in ((sg.halo, emu.halo), (sg.halo_t, emu.halo_t)):
                assert torch.equal(real.send_idx, fake.send_idx) and list(real.send_counts) == list(fake.send_counts)
                assert torch.equal(real.col, fake.col) and list(real.recv_counts) == list(fake.recv_counts)
                assert torch.equal(real.interior, fake.interior) and torch.equal(real.boundary, fake.boundary)
            extra = (sg.stats["exchanged_bytes"], (sg.halo.n_halo + sg.halo_t.n_halo) * fout * (2 if halo_dtype == "bf16" else 4))
        q.put((rank, lo, hi, y.detach().numpy(), wt.grad.numpy(), xl.grad.numpy(), None if bt is None else bt.grad.numpy(), extra))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,use_bias,active,mode", [(2, 101, False, True, "allgather"), (2, 64, True, False, "allgather"),
                                                         (3, 50, True, True, "allgather"), (2, 101, True, True, "alltoall"),
                                                         (3, 50, False, True, "alltoall"),
                                                         # the driver's scaling run goes to 4 and 8 ranks
                                                         (4, 90, True, True, "allgather"), (8, 100, False, True, "alltoall"),
                                                         (8, 37, True, True, "alltoall"),
                                                         # halo exchange (all-to-all-v of the referenced rows only), uneven shards,
                                                         # ranks with nothing to send, bias / no bias, with and without ReLU
                                                         (2, 101, False, True, "halo"), (3, 50, True, True, "halo"),
                                                         (8, 100, True, False, "halo"), (8, 37, False, True, "halo"),
                                                         (1, 40, True, True, "halo")])
def test_sharded_layer_matches_oracle(world, n, use_bias, active, mode):
    fin, fout, k, seed = 12, 24, 5, 17 + n  # fout divisible by 2, 3, 4 and 8 (alltoall mode slices the layer width)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, fin, fout, k, seed, use_bias, active, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, w, b, dy, adj = _problem(n, fin, fout, k, seed)
    ref = ol.gcn_layer_fwd_bwd(x, adj, w, dy, bias=b if use_bias else None, active=active, x_requires_grad=True)
    results.sort(key=lambda r: r[0])
    y = np.concatenate([r[3] for r in results])
    dx = np.concatenate([r[5] for r in results])
    assert y.shape == ref["out"].shape
    assert rel_err(y, ref["out"]) < 1e-5
    assert rel_err(dx, ref["dX"]) < 1e-5
    for r in results:  # all-reduced gradients are identical on every rank
        assert rel_err(r[4], ref["dW"]) < 1e-5
        if use_bias:
            assert rel_err(r[6], ref["db"]) < 1e-5
        if r[7] is not None:
            assert r[7][0] == r[7][1]  # bytes on the wire = referenced remote rows x width x 4


def _halo_fused_worker(rank, world, port, n, fin, fout, k, seed, halo_dtype, q):
    _worker(rank, world, port, n, fin, fout, k, seed, False, True, q, "halo", halo_dtype)


def _halo_linear_bf16_worker(rank, world, port, n, fin, fout, k, seed, q):
    _worker(rank, world, port, n, fin, fout, k, seed, True, False, q, "halo", "bf16")


def test_halo_bf16_wire_linear_layer():
    """bf16 halos on a layer without ReLU (no mask flips): the only difference to one rank is the 2^-9 rounding of the rows
    that crossed the wire."""
    world, n, fin, fout, k, seed = 3, 80, 10, 24, 6, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_linear_bf16_worker, args=(r, world, port, n, fin, fout, k, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, w, b, dy, adj = _problem(n, fin, fout, k, seed)
    ref = ol.gcn_layer_fwd_bwd(x, adj, w, dy, bias=b, active=False, x_requires_grad=True)
    out, dx = np.concatenate([r[3] for r in results]), np.concatenate([r[5] for r in results])
    assert 1e-7 < rel_err(out, ref["out"]) < 1e-2 and rel_err(dx, ref["dX"]) < 1e-2   # rounded, but only at the bf16 level
    for r in results:
        assert rel_err(r[4], ref["dW"]) < 1e-2 and r[7][0] == r[7][1]


@pytest.mark.parametrize("world,n,halo_dtype", [(2, 90, "f32"), (3, 77, "f32"), (8, 120, "f32")])
def test_halo_fused_relu_mask_path(world, n, halo_dtype):
    """Layer width 128: the fused ReLU-mask SpMM pair stays in use at P > 1 — local rows of G are masked inside the backward
    SpMM, halo rows while they are packed (all-ones mask words for the received rows) — and the interior / boundary row split.
    fp32 halos: same sums as one rank."""
    fin, fout, k, seed = 10, 128, 6, 3 + n
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_fused_worker, args=(r, world, port, n, fin, fout, k, seed, halo_dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, w, b, dy, adj = _problem(n, fin, fout, k, seed)
    ref = ol.gcn_layer_fwd_bwd(x, adj, w, dy, bias=None, active=True, x_requires_grad=True)
    tol = 1e-5
    assert rel_err(np.concatenate([r[3] for r in results]), ref["out"]) < tol
    assert rel_err(np.concatenate([r[5] for r in results]), ref["dX"]) < tol
    for r in results:
        assert rel_err(r[4], ref["dW"]) < tol and r[7][0] == r[7][1]


def _scaled_worker(rank, world, port, n, fin, fout, k, seed, reduce, q, mode):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    from dance_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, w, b, dy, adj = _problem(n, fin, fout, k, seed)
        at = adj.T.tocsr()
        at.sort_indices()
        lo, hi = sharding.row_ranges(n, world)[0][rank]
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
        sl = lambda m, a, b_: sharding.slice_rows(t(m.indptr, np.int32), t(m.indices, np.int32), t(m.data, np.float32), a, b_, n)
        full = (sl(adj, 0, n), sl(at, 0, n)) if mode == "alltoall" else None
        sg = sharding.ShardedGCNGraph(sl(adj, lo, hi), sl(at, lo, hi), n, mode=mode, full=full)
        rs, cs = _norm_both(adj)
        xl = torch.from_numpy(x[lo:hi].copy()).requires_grad_(True)
        wt = torch.from_numpy(w.copy()).requires_grad_(True)
        bt = torch.from_numpy(b.copy()).requires_grad_(True)
        y = sharding.sharded_gcn_layer(xl, wt, sg, bt, True, ops=cpu_ops, rowscale=torch.from_numpy(rs),
                                       colscale=torch.from_numpy(cs), reduce=reduce)
        y.backward(torch.from_numpy(dy[lo:hi].copy()))
        q.put((rank, y.detach().numpy(), wt.grad.numpy(), xl.grad.numpy(), bt.grad.numpy()))
    finally:
        dist.destroy_process_group()


def _norm_both(adj):
    """GraphConv(norm="both") scales of graphsc.py:444-476: in-degree^-1/2 per destination, out-degree^-1/2 per source."""
    in_deg = np.maximum(np.diff(adj.indptr), 1).astype(np.float32)
    out_deg = np.maximum(np.bincount(adj.indices, minlength=adj.shape[0]), 1).astype(np.float32)
    return in_deg**-0.5, out_deg**-0.5


@pytest.mark.parametrize("world,n,reduce,mode", [(2, 77, 0, "allgather"), (2, 77, 1, "alltoall"), (3, 50, 1, "allgather"),
                                                (3, 50, 0, "alltoall")])
def test_sharded_graphconv_norm_both_matches_dense_reference(world, n, reduce, mode):
    """The scaled / mean-reduced layer (WeightedGraphConv full-graph form, BASELINE config 4) sharded over `world`
    ranks against a float64 dense autograd reference."""
    fin, fout, k, seed = 10, 6, 5, 5 + n
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scaled_worker, args=(r, world, port, n, fin, fout, k, seed, reduce, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, w, b, dy, adj = _problem(n, fin, fout, k, seed)
    rs, cs = _norm_both(adj)
    a = torch.from_numpy(adj.toarray()).double()
    if reduce == 1:
        a = a / torch.from_numpy(np.maximum(np.diff(adj.indptr), 1)).double()[:, None]
    xt, wt, bt = (torch.from_numpy(v).double().requires_grad_(True) for v in (x, w, b))
    y = torch.relu(torch.from_numpy(rs).double()[:, None] * (a @ (torch.from_numpy(cs).double()[:, None] * (xt @ wt))) + bt)
    y.backward(torch.from_numpy(dy).double())
    assert rel_err(np.concatenate([r[1] for r in results]), y.detach().numpy()) < 1e-5
    assert rel_err(np.concatenate([r[3] for r in results]), xt.grad.numpy()) < 1e-5
    for r in results:
        assert rel_err(r[2], wt.grad.numpy()) < 1e-5 and rel_err(r[4], bt.grad.numpy()) < 1e-5


def _knn_worker(rank, world, port, n, d, k, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    from dance_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.from_numpy(np.random.default_rng(3).standard_normal((n, d)).astype(np.float32))
        idx, dst = sharding.sharded_knn(x, k, ops=cpu_ops)
        q.put((rank, idx.numpy(), dst.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 203), (3, 100)])
def test_sharded_knn_equals_single_process(world, n):
    from oracle import graphs as og
    d, k = 7, 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_knn_worker, args=(r, world, port, n, d, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref_idx, ref_dist = og.knn_exact(np.random.default_rng(3).standard_normal((n, d)).astype(np.float32), k)
    for _, idx, dst in results:  # every rank holds the full, identical result
        assert np.array_equal(idx, ref_idx) and np.array_equal(dst, ref_dist)


def test_row_ranges_cover_everything():
    from dance_amd.sharding import row_ranges
    for n in (0, 1, 7, 8, 9, 1_000_000):
        for world in (1, 2, 3, 8):
            ranges, chunk = row_ranges(n, world)
            assert ranges[0][0] == 0 and ranges[-1][1] == n and len(ranges) == world
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            assert all(hi - lo <= chunk for lo, hi in ranges)


def _dp_worker(rank, world, port, q):
    from dance_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)                       # different initial weights per rank before the broadcast
        model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
        sharding.broadcast_parameters(model)
        ids = torch.arange(10, 10 + 23)                     # 23 seed cells over `world` ranks: uneven
        mine = sharding.shard_seed_ids(ids)
        x = torch.arange(40 * 6, dtype=torch.float32).reshape(40, 6).sin()
        model.train()
        loss = model(x[mine]).pow(2).mean()
        loss.backward()
        sharding.allreduce_gradients(model)
        z, order = sharding.gather_embeddings(x[mine][:, :3].contiguous(), mine)
        q.put((rank, mine.numpy(), [p.grad.numpy().copy() for p in model.parameters()], [p.detach().numpy().copy() for p in model.parameters()],
               z.numpy(), order.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_helpers(world):
    """shard_seed_ids / broadcast_parameters / allreduce_gradients / gather_embeddings (mini-batch data parallelism of
    ScDeepSort and GraphSC, BASELINE config 4): equal batch counts, identical weights, averaged gradients, one row per cell."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    per = -(-23 // world)
    assert all(r[1].size == per for r in res)                                     # every rank runs the same number of steps
    assert all(np.unique(r[1]).size == per for r in res)                          # ... over DISTINCT seeds (dh_block_plan's contract)
    assert set(np.concatenate([r[1] for r in res])) == set(range(10, 33))         # every cell is somebody's
    for r in res[1:]:
        for a, b in zip(r[3], res[0][3]):
            assert np.array_equal(a, b)                                           # broadcast: identical weights
        for a, b in zip(r[2], res[0][2]):
            assert np.array_equal(a, b)                                           # all-reduced: identical gradients
    # the averaged gradient equals the mean of the per-rank gradients of the same model
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
    x = torch.arange(40 * 6, dtype=torch.float32).reshape(40, 6).sin()
    acc = None
    for r in res:
        model.zero_grad()
        model(x[torch.from_numpy(r[1])]).pow(2).mean().backward()
        g = [p.grad.clone() for p in model.parameters()]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    for a, b in zip(acc, res[0][2]):
        assert np.allclose((a / world).numpy(), b, rtol=1e-5, atol=1e-7)
    for r in res:                                                                 # gathered embeddings: one row per cell, sorted
        assert np.array_equal(r[5], np.arange(10, 33)) and np.allclose(r[4], x[10:33, :3].numpy())


def _spagcn_worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    from dance_amd import kernels, sharding
    from dance_amd.modules.spatial.spatial_domain.spagcn import SimpleGCDEC
    for name in cpu_ops.STAND_INS:  # the head's fused Student-t kernels (and whatever else the model calls) on the host
        setattr(kernels, name, getattr(cpu_ops, name))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        n, d = 90, 8
        lab = (np.arange(n) >= n // 2).astype(int)
        x = (np.eye(2)[lab] @ rng.standard_normal((2, d)) * 2 + rng.standard_normal((n, d)) * 0.5).astype(np.float32)
        adj = sp.random(n, n, density=0.08, random_state=1, format="csr", dtype=np.float32)
        adj = (adj + adj.T + sp.eye(n)).tocsr().astype(np.float32)
        adj.data[:] = np.exp(-rng.uniform(0, 2, adj.nnz)).astype(np.float32)
        adj.sort_indices()
        at = adj.T.tocsr()
        at.sort_indices()
        lo, hi = sharding.row_ranges(n, world)[0][rank]
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
        sl = lambda m: sharding.slice_rows(t(m.indptr, np.int32), t(m.indices, np.int32), t(m.data, np.float32), lo, hi, n)
        sg = sharding.ShardedGCNGraph(sl(adj), sl(at), n, mode="halo")
        sg.ops = cpu_ops
        torch.manual_seed(3 + rank)   # different initial weights per rank: the fit broadcasts rank 0's
        np.random.seed(0)
        m = SimpleGCDEC(d, d, device="cpu")
        m.fit(x, sg, lr=0.01, epochs=6, opt="admin", init="kmeans", n_clusters=2, tol=0.0, weight_decay=0)
        z, qq = m.predict(x, sg)
        q.put((rank, qq.detach().numpy(), m.gc.weight.detach().numpy().copy(), m.mu.detach().numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_sharded_spagcn_fit_independent_of_world_size():
    """SimpleGCDEC.fit over a destination-range sharded graph (BASELINE config 5): 3 ranks reproduce 1 rank (global q sums for
    the target distribution, the loss mean over all spots, rank-0 k-means broadcast, gathered predictions)."""
    outs = {}
    for world in (1, 3):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_spagcn_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for r in res[1:]:
            assert np.array_equal(r[1], res[0][1]) and np.allclose(r[2], res[0][2], rtol=1e-6, atol=1e-7)   # every rank: same model, same output
        outs[world] = res[0]
    assert rel_err(outs[3][1], outs[1][1]) < 1e-4 and rel_err(outs[3][2], outs[1][2]) < 1e-4 and rel_err(outs[3][3], outs[1][3]) < 1e-5


def _bench_worker(rank, world, port, q):
    """bench.py's own multi-rank helpers (never run on hardware here: the boxes have one GPU) on gloo / CPU tensors:
    time_steps (barrier + MAX over ranks) and time_exchange_only for every exchange mode."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import bench
    import cpu_ops
    from dance_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, fin, fout, k = 90, 12, 8, 5
        x, w, b, dy, adj = _problem(n, fin, fout, k, 11)
        at = adj.T.tocsr()
        at.sort_indices()
        ranges, _ = sharding.row_ranges(n, world)
        lo, hi = ranges[rank]
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
        out = {}
        for mode in ("halo", "allgather", "alltoall"):
            a_sh = sharding.slice_rows(t(adj.indptr, np.int32), t(adj.indices, np.int32), t(adj.data, np.float32), lo, hi, n)
            at_sh = sharding.slice_rows(t(at.indptr, np.int32), t(at.indices, np.int32), t(at.data, np.float32), lo, hi, n)
            full = None
            if mode == "alltoall":
                full = (sharding.slice_rows(t(adj.indptr, np.int32), t(adj.indices, np.int32), t(adj.data, np.float32), 0, n, n),
                        sharding.slice_rows(t(at.indptr, np.int32), t(at.indices, np.int32), t(at.data, np.float32), 0, n, n))
            sg = sharding.ShardedGCNGraph(a_sh, at_sh, n, mode=mode, full=full)
            xl, wt = torch.from_numpy(x[lo:hi].copy()), torch.from_numpy(w.copy()).requires_grad_(True)
            dyl = torch.from_numpy(dy[lo:hi].copy())

            def step():
                wt.grad = None
                sharding.sharded_gcn_layer(xl, wt, sg, None, True, ops=cpu_ops).backward(dyl)

            class _NoTimer:
                def __enter__(self):
                    return self

                def __exit__(self, *a):
                    return False

            elapsed, _ = bench.time_steps(step, dist.barrier, 2, 1, world, torch.device("cpu"), _NoTimer)
            before = sg.stats["exchanged_bytes"]
            ms, per_step = bench.time_exchange_only(sg, fout, 2, dist.barrier, torch.device("cpu"))
            assert sg.stats["exchanged_bytes"] == before  # the dry exchange leaves the layer's own accounting alone
            out[mode] = (elapsed, ms, per_step)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_bench_multirank_helpers_on_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for mode in ("halo", "allgather", "alltoall"):
        e0, ms0, b0 = res[0][mode]
        e1, ms1, b1 = res[1][mode]
        assert e0 == e1 > 0 and ms0 == ms1 > 0  # MAX over ranks: identical on both
        assert b0 > 0 and b1 > 0


def _model_dp_worker(rank, world, port, q):
    """GraphSC.fit and ScDeepSort.fit, the models' own data-parallel loops, under gloo with the kernel stand-ins."""
    import json
    import sys
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    import test_graphsc_host_logic as gh
    from dance_amd import kernels
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    from dance_amd.modules.single_modality.clustering import graphsc
    for name in cpu_ops.STAND_INS:
        setattr(kernels, name, getattr(cpu_ops, name))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gold = np.load(gh.GOLD)
        g = gh._graph(gold)
        kw = json.loads(str(gold["gsc_kw"]))
        torch.manual_seed(7 + rank)                       # different initial weights per rank: fit broadcasts rank 0's
        m = graphsc.GraphSC(**kw, n_clusters=3, device="cpu")
        m.model.decoder.dropout = 0.0
        m.shuffle_generator = torch.Generator().manual_seed(5)
        m.fit(g, epochs=2, lr=1e-2, batch_size=8)
        gsc = (np.array(m.losses), m.get_latent().copy(), [p.detach().numpy().copy() for p in m.model.parameters()])
        n_cells, n_genes = gold["gsc_x"].shape
        labels = torch.arange(n_cells) % 3
        with tempfile.TemporaryDirectory() as tmp:
            torch.manual_seed(11 + rank)
            sds = ScDeepSort(g.ndata["features"].shape[1], 8, 1, "synthetic", "dp", batch_size=8, device="cpu", save_root=tmp, verbose=False)
            sds.shuffle_generator = torch.Generator().manual_seed(9)
            sds.fit(g, labels, epochs=3, lr=1e-2, val_ratio=0.25)
            prob = sds.predict_proba(g)
            sd = [p.detach().numpy().copy() for p in sds.model.parameters()]
        q.put((rank, gsc, prob, sd))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_model_fit_loops_data_parallel_on_gloo():
    """BASELINE config 4 (and scDeepSort's mini-batch loop) with more than one process: every rank trains on its share of the
    seed cells, gradients are averaged once per step, so all ranks end with bit-identical weights; GraphSC's embeddings are
    gathered in cell order (one row per cell, the same on every rank) and the losses stay finite.  The single-process run of
    the same code is the parity-pinned one (tests/test_graphsc_host_logic.py)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, gsc0, prob0, sd0), (_, gsc1, prob1, sd1) = res
    n_cells = np.load(os.path.join(os.path.dirname(__file__), "golden", "graphsc.npz"))["gsc_x"].shape[0]
    assert all(np.array_equal(a, b) for a, b in zip(gsc0[2], gsc1[2]))       # same weights on both ranks after the fit
    assert gsc0[1].shape[0] == n_cells and np.array_equal(gsc0[1], gsc1[1])  # gathered embeddings: one row per cell, identical
    assert np.isfinite(gsc0[0]).all() and np.isfinite(gsc1[0]).all() and len(gsc0[0]) == len(gsc1[0])
    assert all(np.array_equal(a, b) for a, b in zip(sd0, sd1))
    assert prob0.shape[0] == n_cells and np.array_equal(prob0, prob1)


def _row_shard_worker(rank, world, port, n, k, seed, q):
    """from_row_shard (every rank holds ONLY its rows of A; A^T by the set-up exchange) against from_global_csr-style slicing."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from dance_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _, _, _, _, adj = _problem(n, 4, 4, k, seed)
        at = adj.T.tocsr()
        at.sort_indices()
        ranges, _ = sharding.row_ranges(n, world)
        lo, hi = ranges[rank]
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
        a_sh = sharding.slice_rows(t(adj.indptr, np.int32), t(adj.indices, np.int32), t(adj.data, np.float32), lo, hi, n)
        want = sharding.slice_rows(t(at.indptr, np.int32), t(at.indices, np.int32), t(at.data, np.float32), lo, hi, n)
        out = {}
        for mode in ("allgather", "halo"):
            sg = sharding.ShardedGCNGraph.from_row_shard(a_sh, n, mode=mode)
            ref = sharding.ShardedGCNGraph(a_sh, want, n, mode=mode)
            ok = (torch.equal(sg.at.rowptr, want.rowptr) and torch.equal(sg.at.col, want.col) and torch.equal(sg.at.val, want.val)
                  and (sg.at.lo, sg.at.hi) == (lo, hi))
            if mode == "halo":
                for a, b in ((sg.halo, ref.halo), (sg.halo_t, ref.halo_t)):
                    ok = ok and torch.equal(a.col, b.col) and torch.equal(a.remote_ids, b.remote_ids) and a.recv_counts == b.recv_counts \
                        and torch.equal(a.send_idx, b.send_idx) and a.send_counts == b.send_counts
            out[mode] = bool(ok)
        try:
            sharding.ShardedGCNGraph.from_row_shard(a_sh, n, mode="alltoall")
            out["alltoall_rejected"] = False
        except ValueError:
            out["alltoall_rejected"] = True
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 101), (3, 50), (4, 37), (1, 23)])
def test_from_row_shard_builds_the_transposed_shard_by_exchange(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_row_shard_worker, args=(r, world, port, n, 5, 3 + n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, out in results:
        assert out == {"allgather": True, "halo": True, "alltoall_rejected": True}


def _scdsc_worker(rank, world, port, mode, q, halo_dtype="f32"):
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    from dance_amd import kernels, sharding
    for name in cpu_ops.STAND_INS:  # the dense / elementwise ops of the model on the host (what the cpu_kernels fixture does)
        setattr(kernels, name, getattr(cpu_ops, name))
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scdsc_fit.npz"))
        kw = json.loads(str(g["sf_kw"]))
        n = g["sf_x"].shape[0]
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            m = ScDSC(pretrain_path=os.path.join(tmp, f"ae{rank}.pt"), device="cpu", **kw)
            if rank == 0:  # the other ranks start from their own random weights: the fit broadcasts rank 0's after pre-training
                m.model.load_state_dict({k.split("::", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sf_sd0::")})
            adj = sp.csr_matrix((g["sf_adj_data"], g["sf_adj_indices"], g["sf_adj_indptr"]), shape=(n, n))
            adj.sort_indices()
            at = adj.T.tocsr()
            at.sort_indices()
            lo, hi = sharding.row_ranges(n, world)[0][rank]
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
            sl = lambda mm: sharding.slice_rows(t(mm.indptr, np.int32), t(mm.indices, np.int32), t(mm.data, np.float32), lo, hi, n)
            sg = sharding.ShardedGCNGraph(sl(adj), sl(at), n, mode=mode, halo_dtype=halo_dtype)
            sg.ops = cpu_ops
            torch.manual_seed(10)
            if rank > 0:  # pre-training is replicated; ranks other than 0 run it from different weights — and are overwritten
                with torch.no_grad():
                    for prm in m.model.parameters():
                        prm.add_(0.01)
            m.fit((sg, g["sf_x"], g["sf_counts"], g["sf_n_counts"].astype(np.float64)), g["sf_y"], lr=1e-3, epochs=12, pt_epochs=3,
                  pt_batch_size=32, pt_lr=1e-3)
            sd = {k: v.detach().numpy().copy() for k, v in m.model.state_dict().items()}
            q.put((rank, m.predict_proba(), m.predict(), sd, float(m.last_loss)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,halo_dtype", [(2, "halo", "f32"), (3, "allgather", "f32"), (2, "halo", "bf16")])
def test_sharded_scdsc_fit_vs_reference_golden(world, mode, halo_dtype):
    """ScDSC.fit with the cells sharded by destination range over 2 / 3 ranks (7 chained sharded GCN layers, BatchNorm statistics,
    target distribution and loss means over ALL cells by all-reduce) reproduces the reference's own single-process fit
    (tests/golden/scdsc_fit.npz) to the tolerance of the single-process test; every rank ends with the same model and the same q.
    ``halo_dtype="bf16"``: the rows that cross the wire are rounded to bf16 (2^-9 relative) in every one of the 7 layers and their
    backward — the fit still lands on the reference's clustering, at a looser tolerance."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scdsc_fit.npz"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_scdsc_worker, args=(r, world, port, mode, q, halo_dtype)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res[1:]:
        assert np.array_equal(r[1], res[0][1]) and r[4] == res[0][4]
        for k, v in r[3].items():
            assert np.allclose(v, res[0][3][k], rtol=1e-6, atol=1e-7), k
    qq, pred, sd = res[0][1], res[0][2], res[0][3]
    assert qq.shape == g["sf_q"].shape and np.allclose(qq.sum(1), 1, atol=1e-5)
    loose = halo_dtype == "bf16"
    assert rel_err(qq, g["sf_q"]) < (5e-2 if loose else 5e-3)
    assert (pred == g["sf_pred"]).mean() > (0.95 if loose else 0.98)
    for k in g.files:
        if k.startswith("sf_sd1::") and "num_batches_tracked" not in k:
            assert np.abs(sd[k.split("::", 1)[1]] - g[k]).max() < (5e-2 if loose else 1.5e-2) * max(1.0, np.abs(g[k]).max()), k


def _graphsc_full_worker(rank, world, port, n_layers, hidden_bn, q):
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_ops
    from dance_amd import kernels
    for name in cpu_ops.STAND_INS:
        setattr(kernels, name, getattr(cpu_ops, name))
    from test_graphsc_host_logic import _graph
    from dance_amd.modules.single_modality.clustering import graphsc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graphsc.npz"))
        kw = json.loads(str(gold["gsc_kw"]))
        kw.update(n_layers=n_layers, hidden_bn=hidden_bn)
        torch.manual_seed(7 + rank)  # different initial weights per rank: rank 0's are broadcast
        m = graphsc.GraphSC(**kw, n_clusters=3, device="cpu")
        m.model.decoder.dropout = 0.0
        if rank == 0 and n_layers == 1 and not hidden_bn:
            m.model.load_state_dict({k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("gsc_full_sd0::")})
        elif rank == 0:
            torch.manual_seed(99)
            m = graphsc.GraphSC(**kw, n_clusters=3, device="cpu")
            m.model.decoder.dropout = 0.0
        m.ops = cpu_ops
        m.fit_full_graph(_graph(gold), epochs=int(os.environ.get("GSC_EPOCHS", 3)), lr=1e-2)
        q.put((rank, np.asarray(m.losses), m.get_latent(), {k: v.numpy().copy() for k, v in m.model.state_dict().items()}))
    finally:
        dist.destroy_process_group()


def _run_graphsc_full(world, n_layers, hidden_bn):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graphsc_full_worker, args=(r, world, port, n_layers, hidden_bn, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res[1:]:  # every rank: same losses, same gathered embedding, same model
        assert np.array_equal(r[1], res[0][1]) and np.array_equal(r[2], res[0][2])
        for k, v in r[3].items():
            assert np.allclose(v, res[0][3][k], rtol=1e-6, atol=1e-7), k
    return res[0]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_full_graph_graphsc_vs_reference_golden(world):
    """GraphSC.fit_full_graph with the cells sharded over 2 / 3 ranks and the genes replicated (BASELINE config 4) ==
    the reference's own fit with the whole cell set as one batch (tests/golden/graphsc.npz, tag "full")."""
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graphsc.npz"))
    _, losses, z, sd = _run_graphsc_full(world, 1, False)
    assert np.allclose(losses, gold["gsc_full_losses"], rtol=2e-4, atol=0)
    assert rel_err(z, gold["gsc_full_z"]) < 1e-3
    for k in gold.files:
        if k.startswith("gsc_full_sd1::"):
            assert rel_err(sd[k.split("::", 1)[1]], gold[k]) < 1e-3, k


def test_sharded_full_graph_graphsc_two_layers_independent_of_world_size():
    """Two GraphConv layers: the inner layer's gene rows are partial sums over each rank's cells, summed by the G x D all-reduce (and
    again in the backward).  3 ranks reproduce 1 rank.  (BatchNorm in the encoder is covered by test_sharded_batch_norm: behind a
    BatchNorm the gradient of the preceding biases is exactly zero in exact arithmetic, and Adam turns its rounding noise into
    +-lr steps — a fit-level comparison would compare noise.)"""
    one = _run_graphsc_full(1, 2, False)
    three = _run_graphsc_full(3, 2, False)
    assert np.allclose(three[1], one[1], rtol=1e-5, atol=0)
    assert rel_err(three[2], one[2]) < 1e-5
    for k, v in one[3].items():
        assert rel_err(three[3][k], v) < 1e-5, k


def _bn_worker(rank, world, port, q):
    from dance_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        n, c = 53, 7
        x = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32) * 3 + 1)
        wgt = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32))  # loss = sum(y * wgt): every rank its rows' share
        lo, hi = sharding.row_ranges(n, world)[0][rank]
        bn = torch.nn.BatchNorm1d(c)
        with torch.no_grad():
            bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
            bn.bias.copy_(torch.from_numpy(rng.standard_normal(c).astype(np.float32)))
        xl = x[lo:hi].clone().requires_grad_(True)
        y = sharding.sharded_batch_norm(bn, xl, n)
        (y * wgt[lo:hi]).sum().backward()
        sharding.allreduce_sum_gradients([bn.weight, bn.bias])
        bn.eval()
        ye = sharding.sharded_batch_norm(bn, x[lo:hi], n)   # eval mode: the running statistics, row-local
        q.put((rank, lo, hi, y.detach().numpy(), xl.grad.numpy(), bn.weight.grad.numpy(), bn.bias.grad.numpy(), bn.running_mean.numpy().copy(),
               bn.running_var.numpy().copy(), int(bn.num_batches_tracked), ye.detach().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batch_norm(world):
    """sharding.sharded_batch_norm over row shards == nn.BatchNorm1d over all rows: output, input gradient, parameter gradients (after
    the sum over ranks), running statistics, eval-mode output."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bn_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    n, c = 53, 7
    x = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32) * 3 + 1).requires_grad_(True)
    wgt = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32))
    bn = torch.nn.BatchNorm1d(c)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, c).astype(np.float32)))
        bn.bias.copy_(torch.from_numpy(rng.standard_normal(c).astype(np.float32)))
    y = bn(x)
    (y * wgt).sum().backward()
    bn.eval()
    ye = bn(x.detach())
    for r in res:
        _, lo, hi, yl, dxl, dw, db, rm, rv, nb, yel = r
        assert np.allclose(yl, y.detach().numpy()[lo:hi], rtol=1e-5, atol=1e-6)
        assert np.allclose(dxl, x.grad.numpy()[lo:hi], rtol=1e-4, atol=1e-5)
        assert np.allclose(dw, bn.weight.grad.numpy(), rtol=1e-5, atol=1e-5) and np.allclose(db, bn.bias.grad.numpy(), rtol=1e-5, atol=1e-5)
        assert np.allclose(rm, bn.running_mean.numpy(), rtol=1e-6, atol=1e-7) and np.allclose(rv, bn.running_var.numpy(), rtol=1e-6, atol=1e-7)
        assert nb == 1 and np.allclose(yel, ye.detach().numpy()[lo:hi], rtol=1e-5, atol=1e-6)
