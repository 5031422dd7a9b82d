"""One process per GPU over RCCL (torch.distributed backend "nccl" = RCCL on ROCm): the tests that RUN the day more than one GPU is
visible, and skip themselves on the one-GPU boxes this suite normally gets (nothing in the repo had ever been
written to run at world > 1 on RCCL).  Same workers and the same bit-identity assertions as the ranks-on-one-GPU tests
(tests/test_gpu_sharded_one_gpu.py), with DANCE_TEST_BACKEND=nccl placing rank r on GPU r:

* the sharded GCN layer in every exchange mode (halo all-to-all-v, all-gather, feature-sliced all-to-all), fp32 and bf16 halo rows, with
  and without the locality renumbering: Y bit-identical to one GPU, dW / db to the summation order;
* the same layer through the C ABI's own collectives (DANCE_AMD_TRANSPORT=capi: dh_comm_halo_spmm_f32 = grouped ncclSend / ncclRecv);
* the sharded exact kNN;
* ``ScDSC.fit`` at world 2 against the reference's own single-process fit (tests/golden/scdsc_fit.npz);
* the data-parallel mini-batch fits (GraphSC / ScDeepSort).
"""
import json
import os
import tempfile

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _need(world):
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n < world:
        pytest.skip(f"needs {world} visible GPUs (one RCCL rank per GPU), found {n}")


@pytest.fixture()
def rccl(monkeypatch):
    monkeypatch.setenv("DANCE_TEST_BACKEND", "nccl")
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (see the build notes)


@pytest.mark.parametrize("world,mode,fout,halo_dtype,reorder", [
    (2, "halo", 512, "f32", None), (2, "allgather", 512, "f32", None), (2, "alltoall", 512, "f32", None), (2, "halo", 512, "f32", "rcm"),
    (2, "halo", 512, "bf16", None), (4, "halo", 512, "f32", None), (4, "alltoall", 512, "f32", None), (8, "halo", 512, "f32", None),
    (8, "allgather", 512, "f32", None), (8, "alltoall", 512, "f32", None)])
def test_sharded_layer_one_rank_per_gpu(cuda_device, rccl, world, mode, fout, halo_dtype, reorder):
    _need(world)
    import test_gpu_sharded_one_gpu as one
    n, fin, k, seed = 6001, 256, 9, 11 + world
    rows_ref, y_ref, dw_ref, db_ref, _ = one._run_layer(0, 1, "allgather", n, fin, fout, k, seed)
    parts = one._spawn(world, "layer", (mode, n, fin, fout, k, seed, halo_dtype, reorder))
    seen = torch.zeros(n, dtype=torch.bool)
    tol = 2e-5 if halo_dtype == "f32" else 5e-2
    for rows, y, dwr, dbr, stats in parts:
        seen[rows] = True
        if halo_dtype == "f32":
            assert torch.equal(y, y_ref[rows]), f"{mode}: this rank's rows differ from the single-GPU layer"
        else:
            assert float((y - y_ref[rows]).abs().max()) <= 1e-2 * float(y_ref.abs().max())
        assert torch.equal(dwr, parts[0][2]) and stats["exchanges"] > 0
        assert float((dbr - db_ref).abs().max()) <= tol * float(db_ref.abs().max())
    assert bool(seen.all()) and float((parts[0][2] - dw_ref).abs().max()) <= tol * float(dw_ref.abs().max())


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_layer_through_the_c_abi_collectives(cuda_device, rccl, monkeypatch, world):
    """DANCE_AMD_TRANSPORT=capi: pack -> dh_comm_halo_exchange_f32 || interior rows -> boundary rows, all inside libdancehip.so."""
    _need(world)
    monkeypatch.setenv("DANCE_AMD_TRANSPORT", "capi")
    import test_gpu_sharded_one_gpu as one
    n, fin, fout, k, seed = 6001, 256, 512, 9, 31
    _, y_ref, dw_ref, _, _ = one._run_layer(0, 1, "allgather", n, fin, fout, k, seed)
    for rows, y, dwr, _, _ in one._spawn(world, "layer", ("halo", n, fin, fout, k, seed, "f32", None)):
        assert torch.equal(y, y_ref[rows])
        assert float((dwr - dw_ref).abs().max()) <= 2e-5 * float(dw_ref.abs().max())


def test_sharded_knn_one_rank_per_gpu(cuda_device, rccl):
    _need(2)
    import test_gpu_sharded_one_gpu as one
    from dance_amd import kernels
    n, d, k, seed = 5003, 24, 12, 3
    x = torch.randn((n, d), generator=torch.Generator().manual_seed(seed)).cuda()
    idx_ref, dst_ref = kernels.knn(x, k)
    for idx, dst in one._spawn(2, "knn", (n, d, k, seed)):
        assert torch.equal(idx, idx_ref.cpu()) and torch.equal(dst, dst_ref.cpu())


def _scdsc_worker(rank, world, port, out_dir):
    import scipy.sparse as sp
    import torch.distributed as dist

    from dance_amd import sharding
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scdsc_fit.npz"))
        kw = json.loads(str(g["sf_kw"]))
        n = g["sf_x"].shape[0]
        with tempfile.TemporaryDirectory() as tmp:
            m = ScDSC(pretrain_path=os.path.join(tmp, f"ae{rank}.pt"), device=f"cuda:{rank}", **kw)
            if rank == 0:
                m.model.load_state_dict({k.split("::", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sf_sd0::")})
            adj = sp.csr_matrix((g["sf_adj_data"], g["sf_adj_indices"], g["sf_adj_indptr"]), shape=(n, n))
            adj.sort_indices()
            at = adj.T.tocsr()
            at.sort_indices()
            lo, hi = sharding.row_ranges(n, world)[0][rank]
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(dev)
            sl = lambda mm: sharding.slice_rows(t(mm.indptr, np.int32), t(mm.indices, np.int32), t(mm.data, np.float32), lo, hi, n)
            sg = sharding.ShardedGCNGraph(sl(adj), sl(at), n, mode="halo")
            torch.manual_seed(10)
            m.fit((sg, g["sf_x"], g["sf_counts"], g["sf_n_counts"].astype(np.float64)), g["sf_y"], lr=1e-3, epochs=12, pt_epochs=3, pt_batch_size=32,
                  pt_lr=1e-3)
            torch.save((m.predict_proba(), m.predict(), {k: v.detach().cpu().numpy() for k, v in m.model.state_dict().items()}),
                       os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_scdsc_fit_world2_vs_reference_golden(cuda_device, rccl):
    """ScDSC.fit with the cells sharded over two GPUs reproduces the reference's own single-process fit (scdsc_fit.npz)."""
    _need(2)
    import torch.multiprocessing as mp

    import test_gpu_sharded_one_gpu as one
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scdsc_fit.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_scdsc_worker, args=(2, one._free_port(), tmp), nprocs=2, join=True)
        res = [torch.load(os.path.join(tmp, f"rank{r}.pt"), weights_only=False) for r in range(2)]
    assert np.array_equal(res[0][0], res[1][0])
    q, pred, sd = res[0]
    assert rel_err(q, g["sf_q"]) < 5e-3 and (pred == g["sf_pred"]).mean() > 0.98
    for k in g.files:
        if k.startswith("sf_sd1::") and "num_batches_tracked" not in k:
            assert np.abs(sd[k.split("::", 1)[1]] - g[k]).max() < 1.5e-2 * max(1.0, np.abs(g[k]).max()), k


def test_mini_batch_fits_data_parallel_one_rank_per_gpu(cuda_device, rccl):
    """GraphSC.fit / ScDeepSort.fit, seed cells sharded over two GPUs, one flat gradient all-reduce per step over RCCL."""
    _need(2)
    import test_gpu_sharded_one_gpu as one
    one.test_model_fit_loops_data_parallel_ranks_on_one_gpu(cuda_device)
