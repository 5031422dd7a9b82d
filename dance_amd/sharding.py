"""Destination-range sharding of the GCN layer across the GPUs of one node.

The reference is single-device; this is new.  One process per GPU (``torch.distributed``,
backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Rank p owns the contiguous node range
[r_p, r_{p+1}): those rows of X, of the CSR of A and of the CSR of A^T, and a replica of W.

    forward : S_p = X_p W                      local MFMA GEMM
              S   = all_gather(S_p)            RCCL, N/P*H*4 bytes per rank, one collective
              Y_p = act(A[p-rows] S + b)       local CSR SpMM (global column ids)
    backward: G_p = dY_p * (Y_p > 0);  G = all_gather(G_p)
              dS_p = A^T[p-rows] G             local CSR SpMM on the transposed shard
              dW   = all_reduce(X_p^T dS_p)    4 MB, latency-bound;  db = all_reduce(colsum(G_p))
              dX_p = dS_p W^T                  local (only if X needs grad)

Every rank computes exactly the rows it owns, in the same per-row order as the single-GPU path, so outputs
are bit-identical to the 1-GPU result and independent of P (dW differs only by the all-reduce order).

Three exchange modes (``ShardedGCNGraph.mode``), same inputs / outputs / numerics:

* ``"halo"`` — what BASELINE.json's north_star names: only the rows a rank's shard actually references travel.  At set-up
  every rank finds the distinct remote columns of its rows of A (and of A^T), asks their owners for them once
  (``HaloPlan``: per-peer send / receive index lists, columns renumbered into [own rows | received rows]) and splits its
  rows into INTERIOR rows (all columns local) and BOUNDARY rows.  Per layer: the GEMM writes S_p straight into the head
  of the operand buffer, the requested rows are packed (dh_gather_rows_f32) and exchanged with one
  ``all_to_all_single`` (variable split sizes = all-to-all-v) that runs asynchronously while the interior rows are
  aggregated (dh_spmm_csr[_relu]_rows_f32); the boundary rows follow when the halo has landed.  The ReLU sign mask is
  kept at P > 1: backward turns dY_p into G_p = dY_p * [Y_p > 0] with one streaming pass (dh_relu_mask_apply_f32) that writes
  straight into the head of the operand buffer — where a plain copy would otherwise be needed — before rows are packed.
  Optional: ``halo_dtype="bf16"`` halves the bytes on the wire (halo rows rounded to bf16: not bit-identical any more),
  ``reorder="rcm"`` renumbers the nodes by reverse Cuthill-McKee first so that kNN-like graphs reference mostly local rows.
* ``"allgather"`` — the dense form of the above: the graph itself is sharded by destination range and every
  rank receives all rows of S (resp. G): (P-1)/P * N*H*4 bytes inbound per all-gather.  Right when most remote
  rows are NOT referenced (kNN graphs after locality reordering) — then it degenerates towards a halo all-to-all-v.
* ``"alltoall"`` — for graphs without locality (the rand-k15 headline graph references ~88 % of all remote rows):
  X, Y and all gradients stay sharded by destination range, but the aggregation itself is done feature-sliced.
  S_p [n_p, H] --all_to_all--> S[:, H_q] for ALL rows; rank q aggregates its H/P columns over the whole graph
  (CSR of A replicated: 12 bytes per edge); Y[:, H_q] --all_to_all--> Y_p [n_p, H].  Each all-to-all moves
  (P-1)/P * N*H*4/P bytes inbound: two of them are 4x (P = 8) less traffic than one all-gather, and xGMI's
  point-to-point links carry all P-1 pairwise transfers concurrently.  Per-row summation order is unchanged.

The compute primitives come from an ``ops`` namespace; the default is ``dance_amd.kernels`` (HIP, fails loudly
without a GPU).  The CPU test-suite injects an oracle-backed namespace to exercise the partition + collective
logic under gloo — the product never selects a CPU backend by itself.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import kernels as _hip_kernels


def row_ranges(n: int, world: int):
    """Equal-size contiguous ranges of size ceil(n/world) (the last ones may be short or empty)."""
    chunk = -(-n // world)
    return [(min(r * chunk, n), min((r + 1) * chunk, n)) for r in range(world)], chunk


@dataclass
class GraphShard:
    """Rows [lo, hi) of a CSR matrix with GLOBAL column ids, row pointers rebased to 0."""
    rowptr: torch.Tensor
    col: torch.Tensor
    val: Optional[torch.Tensor]
    lo: int
    hi: int
    n_cols: int

    @property
    def n_rows(self):
        return self.hi - self.lo


def slice_rows(rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], lo: int, hi: int,
               n_cols: int) -> GraphShard:
    """Cut rows [lo, hi) out of a device/host CSR (index plumbing only)."""
    s, e = int(rowptr[lo]), int(rowptr[hi])
    rp = (rowptr[lo:hi + 1] - rowptr[lo]).contiguous()
    return GraphShard(rp, col[s:e].contiguous(), None if val is None else val[s:e].contiguous(), lo, hi, n_cols)


def transpose_shard(a_shard: "GraphShard", ranges, rank: int, world: int, group=None) -> "GraphShard":
    """Rows [lo, hi) of A^T from every rank's rows of A, by ONE set-up exchange (all-to-all-v of (column, row, value) triples by
    column owner) — no rank ever holds the whole graph.  Row j of the result lists the sources i of the stored entries A[i, j] in
    ascending i (ties in A's own order): exactly the rows ``slice_rows(transpose(A))`` would give, bit for bit."""
    lo, hi = ranges[rank]
    dev = a_shard.col.device
    n_local = a_shard.n_rows
    deg = (a_shard.rowptr[1:] - a_shard.rowptr[:-1]).to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(a_shard.lo, a_shard.lo + n_local, device=dev), deg, output_size=a_shard.col.numel())
    cols = a_shard.col.to(torch.int64)
    vals = a_shard.val if a_shard.val is not None else torch.ones(cols.numel(), dtype=torch.float32, device=dev)
    starts = torch.tensor([r[0] for r in ranges], dtype=torch.int64, device=dev)
    owner = torch.searchsorted(starts, cols, right=True) - 1
    order = torch.argsort(owner, stable=True)  # grouped by destination rank; inside a group still row-major (ascending i)
    send_counts = torch.bincount(owner, minlength=world).tolist()
    pack = torch.stack((cols[order], rows[order]), 1).contiguous()  # int64 [m, 2]
    vsend = vals[order].contiguous()
    if world > 1:
        cnt = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(cnt, torch.tensor(send_counts, dtype=torch.int64, device=dev), group=group)
        recv_counts = cnt.tolist()
        got = torch.empty((sum(recv_counts), 2), dtype=torch.int64, device=dev)
        dist.all_to_all_single(got, pack, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
        vgot = torch.empty(sum(recv_counts), dtype=torch.float32, device=dev)
        dist.all_to_all_single(vgot, vsend, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
    else:
        got, vgot = pack, vsend
    # the pieces arrive grouped by SENDING rank = ascending source row range, each piece row-major: a stable sort by column
    # (= the row of A^T) leaves every row's sources ascending
    j = got[:, 0] - lo
    srt = torch.argsort(j, stable=True)
    rowptr = torch.zeros(hi - lo + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(torch.bincount(j, minlength=hi - lo), 0)
    return GraphShard(rowptr.to(torch.int32), got[srt, 1].to(torch.int32).contiguous(),
                      None if a_shard.val is None else vgot[srt].contiguous(), lo, hi, a_shard.n_cols)


@dataclass
class HaloPlan:
    """Halo bookkeeping of one CSR shard (rows of A, or of A^T): which rows this rank receives from / sends to every peer and
    the shard's columns renumbered into the operand buffer [own rows | rows received from rank 0 | rank 1 | ...]."""
    n_local: int
    col: torch.Tensor                 # int32 [nnz] local column ids
    remote_ids: torch.Tensor          # int64 [n_halo] global ids of the received rows (ascending = grouped by owner)
    recv_counts: List[int]            # rows received from each peer
    send_idx: torch.Tensor            # int32 [sum(send_counts)] local row ids to send, grouped by destination peer
    send_counts: List[int]
    interior: torch.Tensor            # int32 row ids whose columns are all local
    boundary: torch.Tensor            # int32 row ids with at least one received column
    cache: dict = field(default_factory=dict)

    @property
    def n_halo(self) -> int:
        return int(self.remote_ids.numel())


def peer_requests_from_global(rowptr: torch.Tensor, col: torch.Tensor, ranges, rank: int):
    """What the OTHER ranks would ask rank ``rank`` for, computed from the whole CSR instead of by a collective: for every
    peer q the distinct columns in this rank's range among q's rows.  Returns (send_idx int32 local row ids grouped by peer,
    send_counts).  Used when one process plans (or emulates, bench.py --emulate-rank) the shard of a rank of a larger world."""
    lo, hi = ranges[rank]
    n = rowptr.numel() - 1
    dev = col.device
    c = col.to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (rowptr[1:] - rowptr[:-1]).to(torch.int64), output_size=c.numel())
    starts = torch.tensor([r[0] for r in ranges], dtype=torch.int64, device=dev)
    owner = torch.searchsorted(starts, rows, right=True) - 1
    sel = (c >= lo) & (c < hi) & (owner != rank)
    key = torch.unique(owner[sel] * n + c[sel])  # ascending: grouped by peer, then by column
    send_idx = (key % n - lo).to(torch.int32).contiguous()
    send_counts = torch.bincount(key // n, minlength=len(ranges)).tolist()
    return send_idx, send_counts


def build_halo_plan(shard: "GraphShard", ranges, chunk: int, rank: int, world: int, group=None, peer_requests=None) -> HaloPlan:
    """Set-up collective (once per graph): every rank tells the owners which of their rows it needs.  ``peer_requests`` =
    (send_idx, send_counts) from ``peer_requests_from_global`` replaces the collective (single-process planning / emulation)."""
    lo, hi = ranges[rank]
    n_local = hi - lo
    col = shard.col.to(torch.int64)
    dev = col.device
    remote = (col < lo) | (col >= hi)
    uniq = torch.unique(col[remote])  # sorted ascending, hence grouped by owner rank
    slot = torch.searchsorted(uniq, col.clamp(min=0)) if uniq.numel() else torch.zeros_like(col)
    col_local = torch.where(remote, n_local + slot, col - lo).to(torch.int32).contiguous()
    bounds = torch.tensor([r[0] for r in ranges] + [ranges[-1][1]], dtype=torch.int64, device=dev)
    cuts = torch.searchsorted(uniq, bounds)
    recv_counts = (cuts[1:] - cuts[:-1]).tolist()
    # rows with a remote column
    rows_of_edges = torch.repeat_interleave(torch.arange(n_local, device=dev), (shard.rowptr[1:] - shard.rowptr[:-1]).to(torch.int64),
                                            output_size=col.numel())
    n_remote = torch.zeros(n_local, dtype=torch.int64, device=dev).index_add_(0, rows_of_edges, remote.to(torch.int64))
    interior = torch.nonzero(n_remote == 0).reshape(-1).to(torch.int32)
    boundary = torch.nonzero(n_remote > 0).reshape(-1).to(torch.int32)
    # ask the owners: counts first, then the row ids (relative to the owner's range)
    owner_lo = torch.repeat_interleave(bounds[:-1], cuts[1:] - cuts[:-1], output_size=uniq.numel())
    want = (uniq - owner_lo).to(torch.int32)
    if peer_requests is not None:
        send_idx, send_counts = peer_requests
    elif world > 1:
        send_counts_t = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_to_all_single(send_counts_t, torch.tensor(recv_counts, dtype=torch.int64, device=dev), group=group)
        send_counts = send_counts_t.tolist()
        send_idx = torch.empty(sum(send_counts), dtype=torch.int32, device=dev)
        dist.all_to_all_single(send_idx, want.contiguous(), output_split_sizes=send_counts, input_split_sizes=recv_counts, group=group)
    else:
        send_counts, send_idx = [0], torch.empty(0, dtype=torch.int32, device=dev)
    return HaloPlan(n_local, col_local, uniq, recv_counts, send_idx, send_counts, interior, boundary)


def rcm_order(graph) -> torch.Tensor:
    """Reverse Cuthill-McKee order (``perm[new] = old``; dance_amd.graph.locality_order): kNN-like graphs renumbered this way
    reference mostly nearby — hence local — rows, which is what makes the halo small."""
    from .graph import locality_order
    return locality_order(graph, "rcm")


def permute_graph(graph, perm: torch.Tensor):
    """P A P^T for ``perm[new] = old`` (CSRGraph.permute: on the device, the order of the edges inside every row kept, so the
    per-row sums of the renumbered graph are those of the original, bit for bit)."""
    return graph.permute(perm)


class ShardedGCNGraph:
    """This rank's destination-range shard of A and of A^T (+ optionally the replicated graph for "alltoall")."""

    def __init__(self, a_shard: GraphShard, at_shard: GraphShard, n_nodes: int, group=None, *, mode: str = "allgather",
                 full: Optional[Tuple[GraphShard, GraphShard]] = None, halo_dtype: str = "f32", perm: Optional[torch.Tensor] = None,
                 emulate: Optional[Tuple[int, int, object]] = None):
        if mode not in ("allgather", "alltoall", "halo"):
            raise ValueError(f"unknown exchange mode {mode!r}")
        if halo_dtype not in ("f32", "bf16"):
            raise ValueError(f"halo_dtype must be 'f32' or 'bf16', got {halo_dtype!r}")
        self.halo_dtype, self.perm = halo_dtype, perm
        self.stats = {"exchanged_bytes": 0, "exchanges": 0}
        # Who carries the exchanges.  "torch" (default): torch.distributed — the process group the launcher initialised ("nccl" IS RCCL
        # on ROCm; "gloo" in the CPU tests).  "capi" (DANCE_AMD_TRANSPORT=capi; fp32 wire, device tensors): the same three collectives
        # through the C ABI (dh_comm_halo_exchange_f32 / _allgather_rows_f32 / _allreduce_f32 of csrc/comm.hip on a communicator
        # bootstrapped over the group) — what a host without torch.distributed calls.  One RCCL underneath either way.
        import os
        self.transport = os.environ.get("DANCE_AMD_TRANSPORT", "torch")
        if self.transport not in ("torch", "capi"):
            raise ValueError(f"DANCE_AMD_TRANSPORT must be 'torch' or 'capi', got {self.transport!r}")
        self._comm = self._comm_stream = None
        if mode == "alltoall" and full is None:
            raise ValueError("mode='alltoall' needs the replicated graph (full=(A, A^T) as GraphShards over all rows)")
        self.a, self.at = a_shard, at_shard
        self.mode, self.full = mode, full
        self.n_nodes = n_nodes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.emulated = emulate is not None
        if emulate is not None:  # (rank, world, global graph): the shard of one rank of a larger world, planned without collectives
            self.rank, self.world = int(emulate[0]), int(emulate[1])
        self.ranges, self.chunk = row_ranges(n_nodes, self.world)
        lo, hi = self.ranges[self.rank]
        if (a_shard.lo, a_shard.hi) != (lo, hi) or (at_shard.lo, at_shard.hi) != (lo, hi):
            raise ValueError(f"rank {self.rank} must own rows [{lo}, {hi})")
        self.halo = self.halo_t = None
        if mode == "halo":
            req = req_t = None
            if emulate is not None:
                g = emulate[2]
                gt = g.transpose()
                req = peer_requests_from_global(g.rowptr, g.col, self.ranges, self.rank)
                req_t = peer_requests_from_global(gt.rowptr, gt.col, self.ranges, self.rank)
            self.halo = build_halo_plan(a_shard, self.ranges, self.chunk, self.rank, self.world, group, req)
            self.halo_t = build_halo_plan(at_shard, self.ranges, self.chunk, self.rank, self.world, group, req_t)

    @classmethod
    def from_global_csr(cls, graph, group=None, *, mode: str = "allgather", halo_dtype: str = "f32",
                        reorder: Optional[str] = None, emulate: Optional[Tuple[int, int]] = None) -> "ShardedGCNGraph":
        """Slice this rank's rows out of a full ``CSRGraph`` (and its transpose) replicated on every rank.
        ``reorder="rcm"`` first renumbers the nodes (``self.perm[new] = old``: callers feed X / read Y in the new order, i.e.
        ``X_new = X[perm]``, and map results back with ``Y[inv]`` where ``inv[perm] = arange``)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        if emulate is not None:  # (rank, world): plan that rank's shard in this single process; the exchanges themselves cannot run
            rank, world = emulate
        perm = None
        if reorder is not None:
            if reorder != "rcm":
                raise ValueError(f"unknown reordering {reorder!r}")
            perm = rcm_order(graph)
            graph = permute_graph(graph, perm)
        ranges, _ = row_ranges(graph.n_rows, world)
        lo, hi = ranges[rank]
        gt = graph.transpose()
        full = None
        if mode == "alltoall":
            full = (GraphShard(graph.rowptr, graph.col, graph.val, 0, graph.n_rows, graph.n_cols),
                    GraphShard(gt.rowptr, gt.col, gt.val, 0, gt.n_rows, gt.n_cols))
        return cls(slice_rows(graph.rowptr, graph.col, graph.val, lo, hi, graph.n_cols),
                   slice_rows(gt.rowptr, gt.col, gt.val, lo, hi, gt.n_cols), graph.n_rows, group, mode=mode, full=full,
                   halo_dtype=halo_dtype, perm=perm, emulate=None if emulate is None else (rank, world, graph))

    @classmethod
    def from_row_shard(cls, a_shard: GraphShard, n_nodes: int, group=None, *, mode: str = "allgather", halo_dtype: str = "f32") -> "ShardedGCNGraph":
        """From this rank's rows of A alone (every rank calls it with its own range): the rows of A^T the backward needs come from
        one set-up exchange (``transpose_shard``), the halo plans from the usual request collective.  No rank ever materialises the
        whole graph — what ``bench.py --gpus N`` and any caller that generates or loads its graph by row range use.  ("alltoall"
        mode replicates the CSR by design and is only available through ``from_global_csr``.)"""
        if mode == "alltoall":
            raise ValueError("mode='alltoall' needs the replicated graph: use from_global_csr")
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ranges, _ = row_ranges(n_nodes, world)
        if (a_shard.lo, a_shard.hi) != ranges[rank]:
            raise ValueError(f"rank {rank} must own rows [{ranges[rank][0]}, {ranges[rank][1]})")
        return cls(a_shard, transpose_shard(a_shard, ranges, rank, world, group), n_nodes, group, mode=mode, halo_dtype=halo_dtype)

    # ---- halo exchange ("halo" mode) ------------------------------------------------------------------------
    def halo_exchange(self, plan: HaloPlan, send_rows: torch.Tensor, recv_into: torch.Tensor):
        """all-to-all-v of packed rows: ``send_rows`` [sum(send_counts), H] grouped by destination peer -> ``recv_into``
        [n_halo, H] grouped by owner.  Returns a handle whose ``wait()`` makes the current stream wait (None at world 1)."""
        if self.world == 1:
            return None
        h = send_rows.shape[1]
        self.stats["exchanges"] += 1
        if self.transport == "capi" and self.halo_dtype == "f32" and send_rows.is_cuda:
            comm, cs = self._capi(send_rows.device)
            cur = torch.cuda.current_stream(send_rows.device)
            cs.wait_stream(cur)  # the packed rows are ready
            with torch.cuda.stream(cs):
                comm.halo_exchange(send_rows, plan.send_counts, recv_into, plan.recv_counts)
            send_rows.record_stream(cs)
            recv_into.record_stream(cs)
            self.stats["exchanged_bytes"] += plan.n_halo * h * 4

            class _OnCommStream:
                def wait(_self):
                    torch.cuda.current_stream(send_rows.device).wait_stream(cs)
            return _OnCommStream()
        if self.halo_dtype == "bf16":
            wire_send = send_rows.to(torch.bfloat16)
            wire_recv = torch.empty((plan.n_halo, h), dtype=torch.bfloat16, device=send_rows.device)
            self.stats["exchanged_bytes"] += plan.n_halo * h * 2
            work = dist.all_to_all_single(wire_recv, wire_send, output_split_sizes=plan.recv_counts, input_split_sizes=plan.send_counts,
                                          group=self.group, async_op=True)

            class _Widen:
                def wait(_self):
                    work.wait()
                    recv_into.copy_(wire_recv)
            return _Widen()
        self.stats["exchanged_bytes"] += plan.n_halo * h * 4
        return dist.all_to_all_single(recv_into, send_rows, output_split_sizes=plan.recv_counts, input_split_sizes=plan.send_counts,
                                      group=self.group, async_op=True)

    def _capi(self, device):
        """(Communicator over this group, its stream), created on first use."""
        if self._comm is None:
            from .comm import Communicator
            self._comm = Communicator.from_torch_distributed(self.group)
            self._comm_stream = torch.cuda.Stream(device=device)
        return self._comm, self._comm_stream

    def halo_vector(self, plan: HaloPlan, v: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """A replicated global per-node vector in the operand-buffer order of ``plan``: [own range | received rows]."""
        if v is None:
            return None
        lo, hi = self.ranges[self.rank]
        return torch.cat((v[lo:hi], v[plan.remote_ids])).contiguous()

    def as_local_graph(self):
        """world == 1: the shard IS the graph — expose it (with its transpose) as a ``CSRGraph``."""
        if getattr(self, "_local_graph", None) is None:
            from .graph import CSRGraph
            g = CSRGraph(self.a.rowptr, self.a.col, self.a.val, self.a.n_rows, self.a.n_cols)
            gt = CSRGraph(self.at.rowptr, self.at.col, self.at.val, self.at.n_rows, self.at.n_cols)
            g._t, gt._t = gt, g
            self._local_graph = g
        return self._local_graph

    # ---- feature-sliced exchange ("alltoall" mode) ---------------------------------------------------------
    def rows_to_columns(self, local: torch.Tensor) -> torch.Tensor:
        """[n_local, H] (my rows, all columns) -> [world*chunk, H/world] (all rows, my column slice)."""
        h = local.shape[1]
        if h % self.world:
            raise ValueError(f"layer width {h} must be divisible by the world size {self.world} in alltoall mode")
        hq = h // self.world
        send = torch.zeros((self.world, self.chunk, hq), dtype=local.dtype, device=local.device)
        send[:, :local.shape[0]] = local.reshape(local.shape[0], self.world, hq).transpose(0, 1)  # pack per destination
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)
        self.stats["exchanges"] += 1
        self.stats["exchanged_bytes"] += (self.world - 1) * self.chunk * hq * send.element_size()
        return recv.reshape(self.world * self.chunk, hq)  # block r = rows of rank r: global row order

    def columns_to_rows(self, cols: torch.Tensor, n_local: int) -> torch.Tensor:
        """Inverse of ``rows_to_columns``: [world*chunk, H/world] -> [n_local, H]."""
        hq = cols.shape[1]
        send = cols.reshape(self.world, self.chunk, hq).contiguous()
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=self.group)  # recv[q] = column slice q of my rows
        self.stats["exchanges"] += 1
        self.stats["exchanged_bytes"] += (self.world - 1) * self.chunk * hq * send.element_size()
        return recv[:, :n_local].transpose(0, 1).reshape(n_local, self.world * hq).contiguous()

    def all_gather_rows(self, local: torch.Tensor) -> torch.Tensor:
        """[n_local, H] per rank -> [world*chunk, H] (rows of rank r at r*chunk; short shards zero-padded)."""
        if self.world == 1:
            return local
        h = local.shape[1]
        if local.shape[0] != self.chunk:
            pad = torch.zeros((self.chunk, h), dtype=local.dtype, device=local.device)
            pad[:local.shape[0]] = local
            local = pad
        out = torch.empty((self.world * self.chunk, h), dtype=local.dtype, device=local.device)
        if self.transport == "capi" and local.is_cuda and local.dtype == torch.float32:
            self._capi(local.device)[0].allgather_rows(local.contiguous(), out)
        else:
            dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        self.stats["exchanges"] += 1
        self.stats["exchanged_bytes"] += (self.world - 1) * self.chunk * h * local.element_size()
        return out

    def inv_in_degree(self) -> torch.Tensor:
        """1 / max(in-degree, 1) of every node (global [N], cached): the mean reduction's factor in backward."""
        if getattr(self, "_inv_deg", None) is None:
            if self.full is not None:
                rp = self.full[0].rowptr
                deg = (rp[1:] - rp[:-1]).to(torch.float32)
            else:
                rp = self.a.rowptr
                local = (rp[1:] - rp[:-1]).to(torch.float32)
                deg = self.all_gather_rows(local[:, None])[:self.n_nodes, 0] if self.world > 1 else local
            self._inv_deg = 1.0 / deg.clamp(min=1)
        return self._inv_deg

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            if self.transport == "capi" and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
                self._capi(t.device)[0].allreduce_(t)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class _ShardedGCNLayerFn(torch.autograd.Function):
    """Rows [lo, hi) of  act(rowscale * reduce_e(val_e * colscale[src] * (X W)[src]) + b).

    ``rowscale`` / ``colscale`` are GLOBAL per-node vectors replicated on every rank (4 MB at 1M nodes): with
    D_in^-1/2 / D_out^-1/2 they give DGL's GraphConv(norm="both") with edge weights (graphsc.py:444-476), with
    ``reduce=mean`` its agg="mean" variant; all None is the plain GCN layer of scDSC / SpaGCN."""

    @staticmethod
    def forward(ctx, x_local, weight, bias, sg: ShardedGCNGraph, active: bool, ops, rowscale, colscale, reduce):
        w = weight.contiguous()
        act = ops.ACT_RELU if active else ops.ACT_NONE
        lo, hi = sg.ranges[sg.rank]
        if sg.mode == "halo":
            return _halo_forward(ctx, x_local, w, bias, sg, active, ops, rowscale, colscale, reduce)
        s_local = ops.gemm(x_local, w)
        if sg.mode == "alltoall" and sg.world > 1:
            s_cols = sg.rows_to_columns(s_local)
            a = sg.full[0]
            hq = s_cols.shape[1]
            b_cols = None if bias is None else bias[sg.rank * hq:(sg.rank + 1) * hq].contiguous()
            y_cols = ops.spmm_csr(a.rowptr, a.col, a.val, s_cols, n_cols=s_cols.shape[0], rowscale=rowscale,
                                  colscale=_pad_to(colscale, s_cols.shape[0]), bias=b_cols, act=act, reduce=reduce,
                                  tag="spmm_csr_f32[fwd]")
            if y_cols.shape[0] != s_cols.shape[0]:  # pad rows so the block layout matches world * chunk
                pad = torch.zeros((s_cols.shape[0], hq), dtype=y_cols.dtype, device=y_cols.device)
                pad[:y_cols.shape[0]] = y_cols
                y_cols = pad
            out = sg.columns_to_rows(y_cols, x_local.shape[0])
        else:
            s_full = sg.all_gather_rows(s_local)
            out = ops.spmm_csr(sg.a.rowptr, sg.a.col, sg.a.val, s_full, n_cols=s_full.shape[0],
                               rowscale=None if rowscale is None else rowscale[lo:hi].contiguous(),
                               colscale=_pad_to(colscale, s_full.shape[0]), bias=bias, act=act, reduce=reduce,
                               tag="spmm_csr_f32[fwd]")
        ctx.sg, ctx.active, ctx.ops, ctx.has_bias = sg, active, ops, bias is not None
        ctx.rowscale, ctx.colscale, ctx.reduce = rowscale, colscale, reduce
        ctx.save_for_backward(x_local, w, out if active else None, None)
        return out

    @staticmethod
    def backward(ctx, dy):
        x_local, w, out, mask = ctx.saved_tensors
        sg, ops = ctx.sg, ctx.ops
        dy = dy.contiguous()
        if sg.mode == "halo":
            return _halo_backward(ctx, dy, x_local, w, out, mask)
        g_local = ops.relu_backward(out, dy) if ctx.active else dy
        dx = dw = db = None
        lo, hi = sg.ranges[sg.rank]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = sg.all_reduce_sum(ops.colsum(g_local))
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            # d(XW)[u] = colscale[u] * sum_{e: u -> v} val_e * m[v] * G[v],  m = rowscale (/ in-degree for mean): the
            # roles of the two scale vectors swap on the transposed graph
            m = ctx.rowscale
            if ctx.reduce == ops.REDUCE_MEAN:
                m = sg.inv_in_degree() if m is None else m * sg.inv_in_degree()
            if sg.mode == "alltoall" and sg.world > 1:
                g_cols = sg.rows_to_columns(g_local)
                at = sg.full[1]
                ds_cols = ops.spmm_csr(at.rowptr, at.col, at.val, g_cols, n_cols=g_cols.shape[0], rowscale=ctx.colscale,
                                       colscale=_pad_to(m, g_cols.shape[0]), tag="spmm_csr_f32[bwd]")
                if ds_cols.shape[0] != g_cols.shape[0]:
                    pad = torch.zeros((g_cols.shape[0], g_cols.shape[1]), dtype=ds_cols.dtype, device=ds_cols.device)
                    pad[:ds_cols.shape[0]] = ds_cols
                    ds_cols = pad
                ds = sg.columns_to_rows(ds_cols, x_local.shape[0])
            else:
                g_full = sg.all_gather_rows(g_local)
                ds = ops.spmm_csr(sg.at.rowptr, sg.at.col, sg.at.val, g_full, n_cols=g_full.shape[0],
                                  rowscale=None if ctx.colscale is None else ctx.colscale[lo:hi].contiguous(),
                                  colscale=_pad_to(m, g_full.shape[0]), tag="spmm_csr_f32[bwd]")
            if ctx.needs_input_grad[1]:
                dw = sg.all_reduce_sum(ops.gemm(x_local, ds, trans_a=True))
            if ctx.needs_input_grad[0]:
                dx = ops.gemm(ds, w, trans_b=True)
        return dx, dw, db, None, None, None, None, None, None


def _run_split(spmm, plan: HaloPlan, work):
    """Interior rows while the halo is in flight, boundary rows once it has landed."""
    if work is None:  # nothing is travelling: one launch over all rows
        spmm(None)
        return
    if plan.interior.numel():
        spmm(plan.interior)
    work.wait()
    if plan.boundary.numel():
        spmm(plan.boundary)


def _halo_forward(ctx, x_local, w, bias, sg, active, ops, rowscale, colscale, reduce):
    plan, plan_t = sg.halo, sg.halo_t
    n_loc, h = plan.n_local, w.shape[1]
    lo, hi = sg.ranges[sg.rank]
    dev = x_local.device
    buf = torch.empty((n_loc + plan.n_halo, h), dtype=torch.float32, device=dev)  # operand: [S_p | halo rows]
    ops.gemm(x_local, w, out=buf[:n_loc])  # the GEMM writes S_p in place: no copy into the exchange buffer
    send = ops.gather_rows(buf[:n_loc], plan.send_idx) if plan.send_idx.numel() else buf[:0]
    work = sg.halo_exchange(plan, send, buf[n_loc:])
    out = torch.empty((n_loc, h), dtype=torch.float32, device=dev)
    act = ops.ACT_RELU if active else ops.ACT_NONE
    row_bytes = ops.relu_mask_bytes(1, h) if hasattr(ops, "relu_mask_bytes") else 0
    fused = (active and bias is None and rowscale is None and colscale is None and reduce == ops.REDUCE_SUM and row_bytes > 0
             and hasattr(ops, "spmm_csr_relu") and hasattr(ops, "relu_mask_apply"))
    mask = None
    if fused:  # sign mask of Y_p: backward turns dY_p into G_p with it before anything is gathered or sent
        mask = torch.empty(n_loc * row_bytes, dtype=torch.uint8, device=dev)
        _run_split(lambda rows: ops.spmm_csr_relu(sg.a.rowptr, plan.col, sg.a.val, buf, n_cols=buf.shape[0], act=act, out_mask=mask,
                                                  out=out, rows=rows, tag="spmm_csr_f32[fwd]"), plan, work)
    else:
        rs = None if rowscale is None else rowscale[lo:hi].contiguous()
        cs = sg.halo_vector(plan, colscale)
        _run_split(lambda rows: ops.spmm_csr(sg.a.rowptr, plan.col, sg.a.val, buf, n_cols=buf.shape[0], rowscale=rs, colscale=cs,
                                             bias=bias, act=act, reduce=reduce, out=out, rows=rows, tag="spmm_csr_f32[fwd]"), plan, work)
    ctx.sg, ctx.active, ctx.ops, ctx.has_bias = sg, active, ops, bias is not None
    ctx.rowscale, ctx.colscale, ctx.reduce = rowscale, colscale, reduce
    ctx.save_for_backward(x_local, w, out if (active and not fused) else None, mask)
    return out


def _halo_backward(ctx, dy, x_local, w, out, mask):
    sg, ops = ctx.sg, ctx.ops
    plan_t = sg.halo_t
    n_loc, h = plan_t.n_local, dy.shape[1]
    lo, hi = sg.ranges[sg.rank]
    dev = dy.device
    dx = dw = db = None
    fused = mask is not None
    g_local = dy if (fused or not ctx.active) else ops.relu_backward(out, dy)
    if ctx.has_bias and ctx.needs_input_grad[2]:
        db = sg.all_reduce_sum(ops.colsum(g_local))
    if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
        buf = torch.empty((n_loc + plan_t.n_halo, h), dtype=torch.float32, device=dev)  # operand: [G_p | halo rows of G]
        if fused:  # G_p = dY_p * [Y_p > 0] from the sign mask, written straight into the head of the operand (this pass replaces
            ops.relu_mask_apply(g_local, mask, out=buf[:n_loc])  # the plain copy; the gather below then needs no mask: 4 requests
        else:                                                    # per neighbour instead of 5, as on one GPU)
            buf[:n_loc].copy_(g_local)
        send = ops.gather_rows(buf[:n_loc], plan_t.send_idx) if plan_t.send_idx.numel() else buf[:0]
        work = sg.halo_exchange(plan_t, send, buf[n_loc:])
        ds = torch.empty((n_loc, h), dtype=torch.float32, device=dev)
        if fused:
            _run_split(lambda rows: ops.spmm_csr(sg.at.rowptr, plan_t.col, sg.at.val, buf, n_cols=buf.shape[0], out=ds, rows=rows,
                                                 tag="spmm_csr_f32[bwd]"), plan_t, work)
        else:
            m = ctx.rowscale
            if ctx.reduce == ops.REDUCE_MEAN:
                m = sg.inv_in_degree() if m is None else m * sg.inv_in_degree()
            rs = None if ctx.colscale is None else ctx.colscale[lo:hi].contiguous()
            cs = sg.halo_vector(plan_t, m)
            _run_split(lambda rows: ops.spmm_csr(sg.at.rowptr, plan_t.col, sg.at.val, buf, n_cols=buf.shape[0], rowscale=rs, colscale=cs,
                                                 out=ds, rows=rows, tag="spmm_csr_f32[bwd]"), plan_t, work)
        if ctx.needs_input_grad[1]:
            dw = sg.all_reduce_sum(ops.gemm(x_local, ds, trans_a=True))
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(ds, w, trans_b=True)
    return dx, dw, db, None, None, None, None, None, None


def _pad_to(v: Optional[torch.Tensor], n: int) -> Optional[torch.Tensor]:
    """A global per-node vector padded with zeros to the world * chunk rows of an exchanged operand."""
    if v is None or v.numel() == n:
        return v
    out = torch.zeros(n, dtype=v.dtype, device=v.device)
    out[:v.numel()] = v
    return out


def sharded_gcn_layer(x_local: torch.Tensor, weight: torch.Tensor, sg: ShardedGCNGraph,
                      bias: Optional[torch.Tensor] = None, active: bool = False, ops=None, *,
                      rowscale: Optional[torch.Tensor] = None, colscale: Optional[torch.Tensor] = None,
                      reduce: int = 0) -> torch.Tensor:
    """This rank's rows of act(rowscale * reduce(A diag(colscale) (X W)) + b); gradients of W / b are all-reduced, dX
    stays row-sharded.  ``rowscale`` / ``colscale``: optional global [N] vectors (replicated); ``reduce``: 0 sum, 1 mean."""
    if sg.world == 1 and ops is None:  # one GPU: the plain layer op (same kernels, plus the fused ReLU mask path)
        from .autograd import gcn_layer
        return gcn_layer(x_local, weight, sg.as_local_graph(), bias, active, rowscale=rowscale, colscale=colscale, reduce=reduce)
    return _ShardedGCNLayerFn.apply(x_local, weight, bias, sg, active, ops or _hip_kernels, rowscale, colscale, reduce)


def sharded_knn(X: torch.Tensor, k: int, group=None, ops=None):
    """Exact kNN of all rows of ``X`` (replicated on every rank) with the QUERIES split by contiguous range across the
    ranks — ``dh_knn_bruteforce_f32`` takes a query range for exactly this — and the per-rank lists all-gathered, so
    every rank ends up with the full [N, k] (idx int32, dist f32) of the single-GPU call, bit for bit (each query's
    list depends only on that query).  The graph builders' one O(N^2) step scales with the GPU count; the 2 N k 4-byte
    gather is negligible."""
    ops = ops or _hip_kernels
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = X.shape[0]
    ranges, chunk = row_ranges(n, world)
    lo, hi = ranges[rank]
    idx_l, dist_l = ops.knn(X, k, lo, hi) if hi > lo else (torch.empty((0, k), dtype=torch.int32, device=X.device),
                                                            torch.empty((0, k), dtype=torch.float32, device=X.device))
    if world == 1:
        return idx_l, dist_l
    pad_i = torch.full((chunk, k), -1, dtype=torch.int32, device=X.device)
    pad_d = torch.full((chunk, k), float("inf"), dtype=torch.float32, device=X.device)
    pad_i[:hi - lo], pad_d[:hi - lo] = idx_l, dist_l
    out_i = torch.empty((world * chunk, k), dtype=torch.int32, device=X.device)
    out_d = torch.empty((world * chunk, k), dtype=torch.float32, device=X.device)
    dist.all_gather_into_tensor(out_i, pad_i, group=group)
    dist.all_gather_into_tensor(out_d, pad_d, group=group)
    return out_i[:n].contiguous(), out_d[:n].contiguous()


# ---- mini-batch data parallelism (scDeepSort / graph-sc, SURVEY.md §8e: "plain data parallel over seed-cell batches with
#      gradient all-reduce of the (small) model") ---------------------------------------------------------------------------
def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_seed_ids(ids: torch.Tensor, group=None) -> torch.Tensor:
    """This rank's share of the seed cells: contiguous slices of ceil(n / P) ids.  A short (last) slice is padded so that every
    rank runs the same number of batches (the all-reduce of every step needs all of them) — with ids from the FRONT of the list,
    i.e. cells of another rank, never with repeats of its own: all ids of a rank stay distinct, so no batch can contain a seed
    twice whatever the shuffle (dh_block_plan requires unique seeds).  The padded cells are visited by two ranks in an epoch;
    ``gather_embeddings`` keeps one copy per cell."""
    rank, world = world_info(group)
    if world == 1:
        return ids
    n = ids.numel()
    per = -(-n // world)
    mine = ids[rank * per:min(n, (rank + 1) * per)]
    short = per - mine.numel()
    if short > 0:  # per <= n, and the front ids belong to rank 0 (a short rank is never rank 0 unless world > n)
        pad = ids[:short] if rank > 0 else ids[n - short:]
        mine = torch.cat((mine, pad))
    return mine.contiguous()


def broadcast_parameters(module: torch.nn.Module, group=None, src: int = 0):
    """Same initial weights (and buffers) everywhere before data-parallel training."""
    _, world = world_info(group)
    if world == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def allreduce_gradients(module: torch.nn.Module, group=None):
    """Average the gradients over the ranks with ONE all-reduce of a flat bucket (the models on this path have a few hundred
    thousand parameters: one latency-bound collective per step instead of one per tensor)."""
    _, world = world_info(group)
    if world == 1:
        return
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= world
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].reshape(g.shape))
        off += g.numel()


def allreduce_sum_gradients(params, group=None):
    """SUM the gradients of ``params`` over the ranks (one flat bucket).  For row-sharded training where every rank's loss is its
    rows' share of ONE global loss (local sum / global count): the gradient of a replicated parameter is the sum of the ranks'."""
    _, world = world_info(group)
    grads = [p.grad for p in params if p.grad is not None]
    if world == 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].reshape(g.shape))
        off += g.numel()


class _ShardedBatchNormFn(torch.autograd.Function):
    """Training-mode BatchNorm1d over rows that are sharded across the ranks: the statistics are those of ALL rows (two small
    all-reduces forward — sum, then centred sum of squares: the two-pass variance torch computes —, one backward).  Returns the
    local rows normalised with the global statistics; the gradients of weight / bias are the LOCAL shares (the caller sums the
    gradients of replicated parameters over the ranks, ``allreduce_sum_gradients``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, n_total, eps, group):
        s = x.sum(0)
        dist.all_reduce(s, group=group)
        mean = s / n_total
        xc = x - mean
        v = (xc * xc).sum(0)
        dist.all_reduce(v, group=group)
        var = v / n_total                      # biased: what normalises (torch.nn.functional.batch_norm, training=True)
        invstd = torch.rsqrt(var + eps)
        xhat = xc * invstd
        ctx.save_for_backward(xhat, weight, invstd)
        ctx.n_total, ctx.group = n_total, group
        ctx.mark_non_differentiable(mean, var)
        y = xhat * weight + bias if weight is not None else xhat
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        xhat, weight, invstd = ctx.saved_tensors
        dxhat = dy * weight if weight is not None else dy
        sums = torch.stack((dxhat.sum(0), (dxhat * xhat).sum(0)))
        dweight = (dy * xhat).sum(0) if weight is not None else None
        dbias = dy.sum(0) if weight is not None else None
        dist.all_reduce(sums, group=ctx.group)
        dx = invstd * (dxhat - (sums[0] + xhat * sums[1]) / ctx.n_total)
        return dx, dweight, dbias, None, None, None


def sharded_batch_norm(bn: torch.nn.BatchNorm1d, x_local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """``bn(x)`` for a row shard of a batch of ``n_total`` rows: identical (to summation order) to the module applied to all rows
    on one device — batch statistics over all rows in training mode, the running statistics updated with them (momentum, unbiased
    variance, ``num_batches_tracked``), the module's own path in eval mode and at world 1."""
    _, world = world_info(group)
    if world == 1 or not (bn.training or not bn.track_running_stats):
        return bn(x_local)
    y, mean, var = _ShardedBatchNormFn.apply(x_local, bn.weight, bn.bias, int(n_total), bn.eps, group)
    if bn.training and bn.track_running_stats:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1 - m).add_(var * (n_total / max(n_total - 1, 1)), alpha=m)
    return y


def gather_embeddings(z: torch.Tensor, order: torch.Tensor, group=None):
    """All ranks' (embedding rows, cell order ids) -> one copy per cell on every rank (padded duplicates dropped, rows sorted
    by order id).  Every rank contributes the same number of rows (``shard_seed_ids`` pads)."""
    _, world = world_info(group)
    if world > 1:
        zs = torch.empty((world * z.shape[0], z.shape[1]), dtype=z.dtype, device=z.device)
        os_ = torch.empty(world * order.shape[0], dtype=order.dtype, device=order.device)
        dist.all_gather_into_tensor(zs, z.contiguous(), group=group)
        dist.all_gather_into_tensor(os_, order.contiguous(), group=group)
        z, order = zs, os_
    order_sorted, idx = torch.sort(order, stable=True)
    first = torch.ones_like(order_sorted, dtype=torch.bool)
    first[1:] = order_sorted[1:] != order_sorted[:-1]
    return z[idx[first]], order_sorted[first]


# ---- bipartite cell - gene graphs (scDeepSort / graph-sc): cells sharded by range, genes replicated ----------------------
class _SumLeadingRows(torch.autograd.Function):
    """rows [0, g) of ``t`` summed over the ranks (the gene rows' partial sums of a cell-sharded aggregation: G x D floats), the other
    rows untouched.  The adjoint of a sum over ranks whose result every rank uses is the sum of the ranks' gradients."""

    @staticmethod
    def forward(ctx, t, g, group):
        ctx.g, ctx.group = g, group
        out = t.clone()
        head = out[:g].contiguous()
        dist.all_reduce(head, group=group)
        out[:g] = head
        return out

    @staticmethod
    def backward(ctx, grad):
        grad = grad.clone()
        head = grad[:ctx.g].contiguous()
        dist.all_reduce(head, group=ctx.group)
        grad[:ctx.g] = head
        return grad, None, None


class ShardedCellGeneGraph:
    """This rank's part of a CellFeatureGraph (nodes: genes [0, G), then cells): ALL gene nodes (a few thousand rows, replicated) and
    the cells [lo, hi) of a contiguous range, with every edge among them.  A cell's in-edges come from genes and its own self loop,
    so its rows are complete locally; a gene's in-edges come from all cells, so a rank holds a PARTIAL sum of every gene row —
    completed by one all-reduce of G x D floats per layer (and one more in its backward).  Gene - gene edges (the genes' self loops)
    are kept on rank 0 only (weight 0 elsewhere) so that the sum over ranks counts them once.  Degrees are those of the whole graph.

    Two message-flow blocks over the local nodes, as the full-neighbour sampler would produce them for the seed set "all cells":
    ``"all"`` — every node is a destination (inner layers of a multi-layer pass), ``"cells"`` — the cells are (the last layer)."""

    def __init__(self, local, n_genes: int, n_cells: int, cell_range, degrees, group=None, ops=None):
        self.local, self.n_genes, self.n_cells = local, int(n_genes), int(n_cells)
        self.lo, self.hi = cell_range
        self.group, self.ops = group, ops
        self.rank, self.world = world_info(group)
        self.in_deg, self.out_deg_all, self.out_deg_cells = degrees  # float32 vectors over the LOCAL nodes, whole-graph counts
        self._blocks = {}

    @classmethod
    def from_global(cls, graph, group=None, ops=None) -> "ShardedCellGeneGraph":
        """From the whole graph (a ``CellGeneGraph`` in the genes-first layout, present on every rank at set-up)."""
        rank, world = world_info(group)
        g = graph.gene_prefix()
        if g < 0:
            raise ValueError("ShardedCellGeneGraph needs the CellFeatureGraph node layout (genes first, then cells)")
        n = graph.number_of_nodes()
        ranges, _ = row_ranges(n - g, world)
        lo, hi = ranges[rank]
        dev = graph.device
        nodes = torch.cat((torch.arange(g, device=dev), g + torch.arange(lo, hi, device=dev)))
        local = graph.subgraph(nodes)
        rp, col = graph.rowptr.to(torch.int64), graph.col.to(torch.int64)
        in_deg = (rp[1:] - rp[:-1]).to(torch.float32)
        out_all = torch.bincount(col, minlength=n).to(torch.float32)
        out_cells = torch.bincount(col[int(rp[g]):], minlength=n).to(torch.float32)   # sources of the cell rows only
        if rank > 0:  # gene <- gene edges (self loops) count once: on rank 0
            lrp = local.rowptr.to(torch.int64)
            head = local.col[:int(lrp[g])].to(torch.int64) < g
            local.val[:int(lrp[g])][head] = 0
        return cls(local, g, n - g, (lo, hi), (in_deg[nodes], out_all[nodes], out_cells[nodes]), group, ops)

    @property
    def n_local_cells(self) -> int:
        return self.hi - self.lo

    def block(self, kind: str):
        """(CSRGraph of the block over the local nodes, in-degree of its destinations, out-degree of its sources) — degrees of the WHOLE
        graph's block, which is what the reference's DGL blocks report for the seed set "all cells"."""
        if kind not in self._blocks:
            from .graph import CSRGraph
            loc, g = self.local, self.n_genes
            n_loc = loc.number_of_nodes()
            if kind == "all":
                csr = CSRGraph(loc.rowptr, loc.col, loc.val, n_loc, n_loc)
                self._blocks[kind] = (csr, self.in_deg, self.out_deg_all)
            elif kind == "cells":
                first = int(loc.rowptr[g])
                csr = CSRGraph((loc.rowptr[g:] - first).contiguous(), loc.col[first:].contiguous(), loc.val[first:].contiguous(), n_loc - g, n_loc)
                self._blocks[kind] = (csr, self.in_deg[g:], self.out_deg_cells)
            else:
                raise ValueError(f"unknown block kind {kind!r}")
        return self._blocks[kind]

    def all_gather_cells(self, local_rows: torch.Tensor) -> torch.Tensor:
        """This rank's cell rows [n_local_cells, d] -> the rows of ALL cells [n_cells, d], in cell order, on every rank."""
        if self.world == 1:
            return local_rows
        ranges, chunk = row_ranges(self.n_cells, self.world)
        d = local_rows.shape[1]
        pad = torch.zeros((chunk, d), dtype=local_rows.dtype, device=local_rows.device)
        pad[:local_rows.shape[0]] = local_rows
        out = torch.empty((self.world * chunk, d), dtype=local_rows.dtype, device=local_rows.device)
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return out[:self.n_cells]  # every range but the last is a whole chunk: the gathered rows are already contiguous in cell order

    def all_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            t = t.clone()
            dist.all_reduce(t, group=self.group)
        return t


def sharded_cellgene_conv(x_local: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], scg: ShardedCellGeneGraph, kind: str, *,
                          norm: str = "both", agg: str = "sum", relu: bool = False) -> torch.Tensor:
    """One weighted GraphConv (graphsc.py:434-484: ``rst = act(D_in^-1/2 . agg(A . (D_out^-1/2 X) W) + b)`` for norm "both"; 1 / D_in on
    the destination side for "right" / "left") over this rank's part of a cell - gene graph.  ``x_local``: features of the local nodes
    (genes, then this rank's cells).  Returns the block's destination rows: all local nodes (``kind="all"``; the gene rows identical on
    every rank) or this rank's cells (``kind="cells"``)."""
    from . import autograd
    ops = scg.ops or _hip_kernels
    csr, in_deg, out_deg = scg.block(kind)
    colscale = rowscale = None
    if norm == "both":
        colscale = out_deg.clamp(min=1).pow(-0.5)
        rowscale = in_deg.clamp(min=1).pow(-0.5)
    elif norm != "none":
        rowscale = 1.0 / in_deg.clamp(min=1)
    if agg == "mean":
        inv = 1.0 / in_deg.clamp(min=1)
        rowscale = inv if rowscale is None else rowscale * inv
    pre = autograd.gcn_layer(x_local, weight, csr, None, False, colscale=colscale, reduce=ops.REDUCE_SUM)
    if kind == "all" and scg.world > 1:
        pre = _SumLeadingRows.apply(pre, scg.n_genes, scg.group)
    if rowscale is not None:
        pre = pre * rowscale[:, None]
    if bias is not None:
        pre = pre + bias
    return torch.relu(pre) if relu else pre


class _ShardedSelfLoopGramBCE(torch.autograd.Function):
    """This rank's share of mean(binary_cross_entropy_with_logits(Z Z^T, I_self, pos_weight=p)) over ALL n x n cell pairs (graphsc.py:
    208-216 with the whole cell set as the batch; the target's ones are the cells' self loops), as a function of the rank's own rows
    z_p: the rows' pass against the all-gathered Z (dh_gram_pairwise_rect_f32) plus the diagonal corrections.  The logits are
    symmetric, so the gradient of the GLOBAL sum with respect to z_p is 2 O_p (+ the diagonal terms): Z is gathered once, no
    gradient travels.  The sum of the ranks' values is the loss."""

    @staticmethod
    def forward(ctx, z, has_self, p, n_total, scg):
        ops = scg.ops or _hip_kernels
        z_all = scg.all_gather_cells(z.contiguous())
        rowloss, o = ops.gram_pairwise_rect(z.contiguous(), z_all.contiguous())
        xe = (z * z).sum(1)
        term = torch.where(has_self, p * torch.nn.functional.softplus(-xe) - torch.nn.functional.softplus(xe), torch.zeros_like(xe))
        ctx.save_for_backward(z, o, xe, has_self)
        ctx.p, ctx.n_total = p, n_total
        return ((rowloss.sum(dtype=torch.float64) + term.sum(dtype=torch.float64)) / float(n_total)**2).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        z, o, xe, has_self = ctx.saved_tensors
        scale = (g / float(ctx.n_total)**2).to(torch.float32)
        sig = torch.sigmoid(xe)
        c = torch.where(has_self, ctx.p * (sig - 1) - sig, torch.zeros_like(xe))
        return scale * 2.0 * (o + c[:, None] * z), None, None, None, None


def sharded_selfloop_gram_bce(z_local: torch.Tensor, has_self: torch.Tensor, pos_weight: float, scg: ShardedCellGeneGraph) -> torch.Tensor:
    return _ShardedSelfLoopGramBCE.apply(z_local, has_self, float(pos_weight), scg.n_cells, scg)
