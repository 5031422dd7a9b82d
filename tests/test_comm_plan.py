"""CPU, no process group: the grouped send / receive plan of csrc/comm.hip (dh_comm_halo_offsets: the host arithmetic behind
dh_comm_halo_exchange_f32 / dh_comm_halo_spmm_f32) against ``sharding.HaloPlan``.  Every rank of a world of 2 .. 8 is planned in this
process (``ShardedGCNGraph(emulate=...)``); the exchange is then SIMULATED with the C function's offsets — message p -> q is
send_p[send_offset_p[q] : + send_rows_p[q]] landing at recv_q[recv_offset_q[p] : + recv_rows_q[p]], which is what one
ncclGroupStart ... ncclSend / ncclRecv per peer ... ncclGroupEnd does — and must leave every rank with exactly the rows its renumbered
columns point at: the sharded SpMM over [own rows | halo] equals the global SpMM."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err


def _offsets(lib, world, rank, send_rows, recv_rows):
    arr = lambda v: (ctypes.c_int64 * world)(*[int(x) for x in v])
    so, ro = (ctypes.c_int64 * world)(), (ctypes.c_int64 * world)()
    ns, nr = ctypes.c_int64(), ctypes.c_int64()
    rc = lib.dh_comm_halo_offsets(world, rank, arr(send_rows), arr(recv_rows), so, ro, ctypes.byref(ns), ctypes.byref(nr))
    return rc, list(so), list(ro), ns.value, nr.value


@pytest.mark.parametrize("world,n,k", [(2, 57, 4), (3, 100, 6), (4, 201, 5), (8, 130, 9)])
def test_grouped_send_recv_plan_moves_the_halo_rows(world, n, k):
    from dance_amd import _lib, sharding
    from dance_amd.graph import CSRGraph
    lib = _lib.load()
    rng = np.random.default_rng(world * 1000 + n)
    adj = sp.random(n, n, density=k / n, random_state=n, format="csr", dtype=np.float32)
    adj.data = rng.uniform(0.1, 1, adj.nnz).astype(np.float32)
    adj.sort_indices()
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
    g = CSRGraph(t(adj.indptr, np.int32), t(adj.indices, np.int32), t(adj.data, np.float32), n, n)
    at = adj.T.tocsr()
    at.sort_indices()
    gt = CSRGraph(t(at.indptr, np.int32), t(at.indices, np.int32), t(at.data, np.float32), n, n)
    g._t, gt._t = gt, g   # (the transpose on the host: CSRGraph.transpose() is a HIP kernel)
    h = 6
    S = rng.standard_normal((n, h)).astype(np.float32)
    ranges, _ = sharding.row_ranges(n, world)
    plans, offs = [], []
    for r in range(world):
        lo, hi = ranges[r]
        sl = lambda m: sharding.slice_rows(t(m.indptr, np.int32), t(m.indices, np.int32), t(m.data, np.float32), lo, hi, n)
        sg = sharding.ShardedGCNGraph(sl(adj), sl(at), n, mode="halo", emulate=(r, world, g))
        plans.append(sg)
        rc, so, ro, ns, nr = _offsets(lib, world, r, sg.halo.send_counts, sg.halo.recv_counts)
        assert rc == 0 and ns == sum(sg.halo.send_counts) == sg.halo.send_idx.numel() and nr == sg.halo.n_halo
        assert so == list(np.concatenate(([0], np.cumsum(sg.halo.send_counts)[:-1]))) and ro == list(np.concatenate(([0], np.cumsum(sg.halo.recv_counts)[:-1])))
        offs.append((so, ro))
    # the packed send buffers (dh_gather_rows_f32 of the own rows) and the simulated exchange
    send = [S[ranges[r][0]:ranges[r][1]][plans[r].halo.send_idx.numpy().astype(np.int64)] for r in range(world)]
    recv = [np.full((plans[r].halo.n_halo, h), np.nan, np.float32) for r in range(world)]
    for p in range(world):
        for q in range(world):
            cnt = plans[p].halo.send_counts[q]
            assert cnt == plans[q].halo.recv_counts[p]           # a send has its matching receive, same size
            if cnt:
                recv[q][offs[q][1][p]:offs[q][1][p] + cnt] = send[p][offs[p][0][q]:offs[p][0][q] + cnt]
    ref = adj @ S
    for r in range(world):
        lo, hi = ranges[r]
        pl = plans[r].halo
        assert not np.isnan(recv[r]).any()
        assert np.array_equal(recv[r], S[pl.remote_ids.numpy()])   # the halo: the referenced remote rows, grouped by owner, ascending
        operand = np.vstack((S[lo:hi], recv[r]))
        local = sp.csr_matrix((plans[r].a.val.numpy(), pl.col.numpy(), plans[r].a.rowptr.numpy()), shape=(hi - lo, operand.shape[0]))
        assert rel_err(local @ operand, ref[lo:hi]) < 1e-6
        assert sorted(pl.interior.tolist() + pl.boundary.tolist()) == list(range(hi - lo))


def test_plan_rejects_bad_counts():
    from dance_amd import _lib
    lib = _lib.load()
    assert _offsets(lib, 3, 1, [2, 0, 5], [1, 0, 0])[0] == 0
    assert _offsets(lib, 3, 1, [2, 1, 5], [1, 0, 0])[0] != 0     # rows for itself
    assert _offsets(lib, 3, 1, [2, 0, -1], [1, 0, 0])[0] != 0    # negative
    assert _offsets(lib, 3, 3, [0, 0, 0], [0, 0, 0])[0] != 0     # rank outside the world
