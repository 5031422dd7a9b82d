"""The P > 1 code path of dance_amd/sharding.py with the REAL kernels, streams and events: P ranks on one GPU, gloo as the
transport (RCCL refuses two ranks on one device; an 8-GPU node is not available to the test suite).  Every exchange mode must
reproduce the single-GPU layer: Y bit for bit (each row is accumulated in the same CSR order whoever owns it), dW to the
summation order of the per-rank partial products."""
import os
import socket
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(n, fin, fout, k, seed):
    g = torch.Generator().manual_seed(seed)
    # k distinct, non-self, sorted columns per row
    col = torch.rand((n, n - 1), generator=g).topk(k, dim=1).indices
    col = col + (col >= torch.arange(n).unsqueeze(1)).to(col.dtype)
    col = col.sort(dim=1).values.to(torch.int32).reshape(-1)
    rowptr = torch.arange(0, n * k + 1, k, dtype=torch.int32)
    val = torch.rand(n * k, generator=g) * 0.9 + 0.1
    x = torch.randn((n, fin), generator=g)
    w = torch.randn((fin, fout), generator=g) / fin**0.5
    b = torch.randn(fout, generator=g)
    dy = torch.randn((n, fout), generator=g)
    return rowptr, col, val, x, w, b, dy


def _device_index(rank):
    """One GPU for every rank (this file), or one GPU per rank (tests/test_gpu_rccl_multi.py sets DANCE_TEST_BACKEND=nccl)."""
    return rank if os.environ.get("DANCE_TEST_BACKEND", "gloo") == "nccl" else 0


def _run_layer(rank, world, mode, n, fin, fout, k, seed, halo_dtype="f32", reorder=None):
    from dance_amd import sharding
    from dance_amd.graph import CSRGraph
    dev = torch.device("cuda", _device_index(rank) if world > 1 else 0)
    rowptr, col, val, x, w, b, dy = _problem(n, fin, fout, k, seed)
    graph = CSRGraph(rowptr.to(dev), col.to(dev), val.to(dev), n, n)
    sg = sharding.ShardedGCNGraph.from_global_csr(graph, mode=mode, halo_dtype=halo_dtype, reorder=reorder)
    lo, hi = sharding.row_ranges(n, world)[0][rank]
    rows = torch.arange(lo, hi) if sg.perm is None else sg.perm.cpu()[lo:hi].long()  # reordered: this rank owns rows perm[lo:hi] of the input
    xl = x[rows].to(dev)
    wt = w.to(dev).requires_grad_(True)
    bt = b.to(dev).requires_grad_(True)
    y = sharding.sharded_gcn_layer(xl, wt, sg, bt, True)
    y.backward(dy[rows].to(dev))
    torch.cuda.synchronize()
    return rows, y.detach().cpu(), wt.grad.cpu(), bt.grad.cpu(), dict(sg.stats)


def _worker(rank, world, port, what, args, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    backend = os.environ.get("DANCE_TEST_BACKEND", "gloo")
    torch.cuda.set_device(_device_index(rank))
    dist.init_process_group(backend, rank=rank, world_size=world,
                            **({"device_id": torch.device("cuda", _device_index(rank))} if backend == "nccl" else {}))
    try:
        if what == "layer":
            out = _run_layer(rank, world, *args)
        else:
            from dance_amd import sharding
            n, d, k, seed = args
            x = torch.randn((n, d), generator=torch.Generator().manual_seed(seed)).cuda(_device_index(rank))
            idx, dst = sharding.sharded_knn(x, k)
            out = (idx.cpu(), dst.cpu())
        torch.save(out, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _spawn(world, what, args):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(world, _free_port(), what, args, tmp), nprocs=world, join=True)
        return [torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(world)]


@pytest.mark.parametrize("world,mode,fout,halo_dtype,reorder", [
    (2, "halo", 512, "f32", None), (2, "allgather", 512, "f32", None), (2, "alltoall", 512, "f32", None),
    (3, "halo", 384, "f32", None), (3, "alltoall", 384, "f32", None),
    (2, "halo", 512, "f32", "rcm"),   # renumbered by locality: each rank owns rows perm[lo:hi]
    (2, "halo", 512, "bf16", None),   # halo rows travel as bf16: Y within bf16 rounding of the gathered operand
])
def test_sharded_layer_ranks_on_one_gpu(cuda_device, world, mode, fout, halo_dtype, reorder):
    n, fin, k, seed = 6001, 256, 9, 11 + world
    rows_ref, y_ref, dw_ref, db_ref, _ = _run_layer(0, 1, "allgather", n, fin, fout, k, seed)
    assert torch.equal(rows_ref, torch.arange(n))
    parts = _spawn(world, "layer", (mode, n, fin, fout, k, seed, halo_dtype, reorder))
    dw = None
    seen = torch.zeros(n, dtype=torch.bool)
    for rows, y, dwr, dbr, stats in parts:
        seen[rows] = True
        if halo_dtype == "f32":
            assert torch.equal(y, y_ref[rows]), f"{mode}: this rank's rows differ from the single-GPU layer"
        else:
            assert float((y - y_ref[rows]).abs().max()) <= 1e-2 * float(y_ref.abs().max())
        dw = dwr if dw is None else dw  # every rank holds the all-reduced gradient
        assert torch.equal(dwr, dw) and stats["exchanges"] > 0
        # bf16 halo rows flip the ReLU sign of the few outputs that sit within rounding of zero: whole dY entries enter or leave the sums
        tol = 2e-5 if halo_dtype == "f32" else 5e-2
        assert float((dbr - db_ref).abs().max()) <= tol * float(db_ref.abs().max())
    assert bool(seen.all())
    assert float((dw - dw_ref).abs().max()) <= tol * float(dw_ref.abs().max())


def test_sharded_knn_ranks_on_one_gpu(cuda_device):
    """Queries split over 3 ranks, lists all-gathered: identical to the single-GPU search, indices and distances."""
    from dance_amd import kernels
    n, d, k, seed = 5003, 24, 12, 3
    x = torch.randn((n, d), generator=torch.Generator().manual_seed(seed)).cuda()
    idx_ref, dst_ref = kernels.knn(x, k)
    for idx, dst in _spawn(3, "knn", (n, d, k, seed)):
        assert torch.equal(idx, idx_ref.cpu()) and torch.equal(dst, dst_ref.cpu())


def _cell_gene_graph(n_cells, n_genes, per, d, seed):
    from dance_amd import kernels
    from dance_amd.cellgraph import CellGeneGraph
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(seed)
    col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32).reshape(-1)
    rp_x = torch.arange(0, n_cells * per + 1, per, dtype=torch.int32, device=dev)
    val_x = torch.rand(n_cells * per, device=dev, generator=g) + 0.5
    rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col, val_x, n_cells, n_genes)
    rowptr, gcol, gval, eid = kernels.cellgene_graph_assemble(rp_x, col, kernels.csr_row_normalize(rp_x, val_x), rp_t, col_t,
                                                              kernels.csr_row_normalize(rp_t, val_t), perm_t, n_cells, n_genes)
    n_nodes = n_cells + n_genes
    cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
    fid = torch.cat((-torch.ones(n_genes, dtype=torch.int32), torch.arange(n_cells, dtype=torch.int32))).to(dev)
    return CellGeneGraph(rowptr, gcol, gval, eid, n_nodes, {"cell_id": cid, "feat_id": fid, "features": torch.randn(n_nodes, d, device=dev, generator=g)})


def _model_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    backend = os.environ.get("DANCE_TEST_BACKEND", "gloo")
    torch.cuda.set_device(_device_index(rank))
    dist.init_process_group(backend, rank=rank, world_size=world,
                            **({"device_id": torch.device("cuda", _device_index(rank))} if backend == "nccl" else {}))
    try:
        n_cells, n_genes, d = 3000, 200, 50
        cg = _cell_gene_graph(n_cells, n_genes, 20, d, seed=1)
        torch.manual_seed(7 + rank)  # different initial weights per rank: fit broadcasts rank 0's
        m = GraphSC(in_feats=d, n_clusters=4, device="cuda")
        m.shuffle_generator = torch.Generator().manual_seed(5)
        m.fit(cg, epochs=2, batch_size=256)
        gsc = (torch.tensor(m.losses), torch.from_numpy(m.get_latent().copy()), [p.detach().cpu() for p in m.model.parameters()])
        labels = torch.arange(n_cells) % 5
        with tempfile.TemporaryDirectory() as tmp:
            torch.manual_seed(11 + rank)
            sds = ScDeepSort(d, 16, 1, "synthetic", "dp", batch_size=256, device="cuda", save_root=tmp, verbose=False)
            sds.shuffle_generator = torch.Generator().manual_seed(9)
            sds.fit(cg, labels, epochs=2, lr=1e-2, val_ratio=0.25)
            sd = [p.detach().cpu() for p in sds.model.parameters()]
        torch.save((gsc, sd), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_model_fit_loops_data_parallel_ranks_on_one_gpu(cuda_device):
    """GraphSC.fit / ScDeepSort.fit with two ranks (BASELINE config 4's data-parallel loop) on the real kernels: each rank trains on its
    share of the seed cells, one gradient all-reduce per step — both ranks end with bit-identical weights, the gathered embedding has one
    row per cell and is the same on both, losses are finite."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_model_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
        (gsc0, sd0), (gsc1, sd1) = [torch.load(os.path.join(tmp, f"rank{r}.pt")) for r in range(2)]
    assert all(torch.equal(a, b) for a, b in zip(gsc0[2], gsc1[2])) and all(torch.equal(a, b) for a, b in zip(sd0, sd1))
    assert gsc0[1].shape[0] == 3000 and torch.equal(gsc0[1], gsc1[1])
    assert bool(torch.isfinite(gsc0[0]).all()) and bool(torch.isfinite(gsc1[0]).all()) and len(gsc0[0]) == len(gsc1[0])
    assert all(bool(torch.isfinite(p).all()) for p in sd0)
