"""RCCL smoke on the one GPU a gpurun box has: a world-size-1 "nccl" group exercises every collective the sharded
layer issues (dance_amd/sharding.py) with device tensors of the shapes it uses — API misuse (non-contiguous buffers,
wrong split sizes, dtype) fails here instead of on the 8-GPU node.  The numerics of the sharded layer for world > 1
are covered on CPU by tests/test_sharding_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture()
def nccl_world1(cuda_device):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda_device)
    yield cuda_device
    dist.destroy_process_group()


def test_collectives_and_sharded_layer_world1(nccl_world1):
    from dance_amd import autograd, sharding
    from dance_amd.graph import CSRGraph
    dev = nccl_world1
    n, f, h, k = 3000, 96, 256, 7
    g = torch.Generator(device="cpu").manual_seed(0)
    col = torch.randint(0, n, (n, k), generator=g).sort(dim=1).values.reshape(-1).to(torch.int32).to(dev)
    rowptr = (torch.arange(n + 1, dtype=torch.int32) * k).to(dev)
    val = torch.rand(n * k, generator=g).to(dev)
    graph = CSRGraph(rowptr, col, val, n, n)
    x = torch.randn(n, f, generator=g).to(dev)
    w = (torch.randn(f, h, generator=g) * 0.1).to(dev).requires_grad_(True)
    dy = torch.randn(n, h, generator=g).to(dev)

    y_ref = autograd.gcn_layer(x, w, graph, None, True)
    y_ref.backward(dy)
    dw_ref = w.grad.clone()

    for mode in ("allgather", "alltoall", "halo"):
        sg = sharding.ShardedGCNGraph.from_global_csr(graph, mode=mode)
        assert sg.world == 1
        # the raw exchanges on device buffers
        s = torch.randn(n, h, device=dev)
        cols = sg.rows_to_columns(s)                       # all_to_all_single
        assert torch.equal(sg.columns_to_rows(cols, n), s)
        out = torch.empty((sg.world * sg.chunk, h), device=dev)
        dist.all_gather_into_tensor(out, s.contiguous())   # what all_gather_rows issues for world > 1
        assert torch.equal(out[:n], s)
        t = s[:4].clone()
        assert torch.equal(sg.all_reduce_sum(t), s[:4])
        # the sharded autograd function itself (collective branches are skipped at world 1, kernels are not)
        w.grad = None
        y = sharding._ShardedGCNLayerFn.apply(x, w, None, sg, True, sharding._hip_kernels, None, None, 0)
        y.backward(dy)
        assert torch.equal(y, y_ref)
        assert np.allclose(w.grad.cpu().numpy(), dw_ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    # GraphConv(norm="both", agg="mean") form: global scale vectors + mean reduction through the sharded function
    rs, cs = torch.rand(n, device=dev) + 0.5, torch.rand(n, device=dev) + 0.5
    b = torch.randn(h, device=dev).requires_grad_(True)
    w.grad = None
    y_ref = autograd.gcn_layer(x, w, graph, b, True, rowscale=rs, colscale=cs, reduce=1)
    y_ref.backward(dy)
    dw_ref, db_ref = w.grad.clone(), b.grad.clone()
    for mode in ("allgather", "alltoall", "halo"):
        sg = sharding.ShardedGCNGraph.from_global_csr(graph, mode=mode)
        w.grad = b.grad = None
        y = sharding._ShardedGCNLayerFn.apply(x, w, b, sg, True, sharding._hip_kernels, rs, cs, 1)
        y.backward(dy)
        assert rel_err(y.detach().cpu().numpy(), y_ref.detach().cpu().numpy()) < 1e-6
        # (x * (1 / deg) here vs x / deg in gcn_layer: last-bit differences in the mean factor)
        assert rel_err(w.grad.cpu().numpy(), dw_ref.cpu().numpy()) < 1e-5
        assert rel_err(b.grad.cpu().numpy(), db_ref.cpu().numpy()) < 1e-5
    # the all-to-all-v of the halo exchange with explicit (here: trivial) split sizes on device buffers
    sg = sharding.ShardedGCNGraph.from_global_csr(graph, mode="halo")
    assert sg.halo.n_halo == 0 and sg.halo.boundary.numel() == 0 and sg.halo.interior.numel() == n
    send, recv = torch.randn(17, h, device=dev), torch.empty(17, h, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=[17], input_split_sizes=[17], async_op=True).wait()
    assert torch.equal(recv, send)
    # RCM renumbering: the permuted problem gives the permuted result
    sgp = sharding.ShardedGCNGraph.from_global_csr(graph, mode="halo", reorder="rcm")
    perm = sgp.perm.to(dev)
    w.grad = None
    yp = sharding._ShardedGCNLayerFn.apply(x[perm].contiguous(), w, None, sgp, True, sharding._hip_kernels, None, None, 0)
    y0 = autograd.gcn_layer(x, w, graph, None, True)
    assert rel_err(yp.detach().cpu().numpy(), y0[perm].detach().cpu().numpy()) < 1e-5
    # query-sharded kNN through the same group (world 1: the local range is everything)
    from dance_amd import kernels
    pts = torch.randn(5000, 20, device=dev)
    i1, d1 = sharding.sharded_knn(pts, 9)
    i0, d0 = kernels.knn(pts, 9)
    assert torch.equal(i1, i0) and torch.equal(d1, d0)
    dist.barrier()
    torch.cuda.synchronize()


def test_spagcn_sharded_equals_single_process_world1(nccl_world1):
    """BASELINE config 5 plumbing: SpaGCN trained through the destination-range sharded layer (every all-reduce / gather /
    broadcast of the sharded fit issued on a world-size-1 RCCL group) reproduces the plain single-GPU fit."""
    from dance_amd import kernels, sharding
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.spatial.spatial_domain.spagcn import SpaGCN
    dev = nccl_world1
    rng = np.random.default_rng(0)
    side = 30
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    n = xy.shape[0]
    dom = (xy[:, 0] >= side / 2).astype(int)
    embed = (np.eye(2)[dom] @ rng.standard_normal((2, 12)) * 2 + rng.standard_normal((n, 12))).astype(np.float32)
    idx, dist_ = kernels.knn(torch.from_numpy(xy).to(dev), 25)
    order = torch.argsort(idx, dim=1)
    rowptr = torch.arange(0, n * 25 + 1, 25, dtype=torch.int32, device=dev)
    g = CSRGraph(rowptr, torch.gather(idx, 1, order).reshape(-1).contiguous(), torch.gather(dist_, 1, order).reshape(-1).contiguous(), n, n)
    outs = []
    for adj in (g, sharding.ShardedGCNGraph.from_global_csr(g, mode="halo")):
        torch.manual_seed(0)
        np.random.seed(0)
        m = SpaGCN(l=1.5, device="cuda")
        m.fit((embed, adj), init="kmeans", n_clusters=2, epochs=15, lr=0.01, tol=0.0)
        outs.append((m.predict_proba((embed, adj)).cpu().numpy(), m.model.gc.weight.detach().cpu().numpy()))
    assert rel_err(outs[1][0], outs[0][0]) < 1e-4 and rel_err(outs[1][1], outs[0][1]) < 1e-4


def test_dh_comm_entry_points_world1(cuda_device):
    """The C-ABI communicator (dh_comm_*: RCCL bound by dlopen, no torch.distributed) at world size 1: bootstrap from a unique
    id, every collective on device buffers of the layer's shapes, and the halo-overlapped SpMM against the plain kernel (with no
    peers the halo is empty: interior + boundary rows must reproduce dh_spmm_csr_f32 bit for bit, masked send path included)."""
    from dance_amd import kernels
    from dance_amd.comm import UNIQUE_ID_BYTES, Communicator
    dev = cuda_device
    uid = Communicator.unique_id()
    assert len(uid) == UNIQUE_ID_BYTES and any(uid)
    comm = Communicator(1, 0, uid)
    assert (comm.world, comm.rank) == (1, 0)
    n, h, k = 5000, 256, 9
    g = torch.Generator(device="cpu").manual_seed(1)
    s = torch.randn(n, h, generator=g).to(dev)
    assert torch.equal(comm.allgather_rows(s), s)
    t = s[:16].clone()
    assert torch.equal(comm.allreduce_(t), s[:16])
    empty = torch.empty((0, h), device=dev)
    comm.halo_exchange(empty, [0], empty, [0])
    with pytest.raises(Exception):
        comm.halo_exchange(s[:3].contiguous(), [3], torch.empty((3, h), device=dev), [3])  # no exchange with oneself
    with pytest.raises(ValueError):
        comm.halo_exchange(empty, [0, 0], empty, [0, 0])
    col = torch.randint(0, n, (n, k), generator=g).sort(dim=1).values.reshape(-1).to(torch.int32).to(dev)
    rowptr = (torch.arange(n + 1, dtype=torch.int32) * k).to(dev)
    val = torch.rand(n * k, generator=g).to(dev)
    bias = torch.randn(h, generator=g).to(dev)
    perm = torch.randperm(n, generator=g).to(torch.int32).to(dev)
    interior, boundary = perm[: n // 3].contiguous(), perm[n // 3:].contiguous()
    y = comm.halo_spmm(rowptr, col, val, s, n, torch.empty(0, dtype=torch.int32, device=dev), [0], [0], interior, boundary, bias=bias,
                       act=kernels.ACT_RELU)
    torch.cuda.synchronize()
    assert torch.equal(y, kernels.spmm_csr(rowptr, col, val, s, bias=bias, act=kernels.ACT_RELU))
    with pytest.raises(Exception):  # receive counts must add up to the halo rows of the operand
        comm.halo_spmm(rowptr, col, val, s, n - 5, torch.empty(0, dtype=torch.int32, device=dev), [0], [0], interior, boundary)
    comm.close()
    comm.close()  # idempotent
    c2 = Communicator.single()
    assert torch.equal(c2.allgather_rows(s[:8].contiguous()), s[:8])
