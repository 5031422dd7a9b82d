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

The compute primitives come from an ``ops`` namespace; the default is ``dance_amd.kernels`` (HIP, fails loudly
without a GPU).  The CPU test-suite injects an oracle-backed namespace to exercise the partition + collective
logic under gloo — the product never selects a CPU backend by itself.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

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


class ShardedGCNGraph:
    """This rank's destination-range shard of A and of A^T."""

    def __init__(self, a_shard: GraphShard, at_shard: GraphShard, n_nodes: int, group=None):
        self.a, self.at = a_shard, at_shard
        self.n_nodes = n_nodes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ranges, self.chunk = row_ranges(n_nodes, self.world)
        lo, hi = self.ranges[self.rank]
        if (a_shard.lo, a_shard.hi) != (lo, hi) or (at_shard.lo, at_shard.hi) != (lo, hi):
            raise ValueError(f"rank {self.rank} must own rows [{lo}, {hi})")

    @classmethod
    def from_global_csr(cls, graph, group=None) -> "ShardedGCNGraph":
        """Slice this rank's rows out of a full ``CSRGraph`` (and its transpose) replicated on every rank."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        ranges, _ = row_ranges(graph.n_rows, world)
        lo, hi = ranges[rank]
        gt = graph.transpose()
        return cls(slice_rows(graph.rowptr, graph.col, graph.val, lo, hi, graph.n_cols),
                   slice_rows(gt.rowptr, gt.col, gt.val, lo, hi, gt.n_cols), graph.n_rows, group)

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
        dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


class _ShardedGCNLayerFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x_local, weight, bias, sg: ShardedGCNGraph, active: bool, ops):
        w = weight.contiguous()
        s_local = ops.gemm(x_local, w)
        s_full = sg.all_gather_rows(s_local)
        out = ops.spmm_csr(sg.a.rowptr, sg.a.col, sg.a.val, s_full, n_cols=s_full.shape[0], bias=bias,
                           act=ops.ACT_RELU if active else ops.ACT_NONE, tag="spmm_csr_f32[fwd]")
        ctx.sg, ctx.active, ctx.ops, ctx.has_bias = sg, active, ops, bias is not None
        ctx.save_for_backward(x_local, w, out if active else None)
        return out

    @staticmethod
    def backward(ctx, dy):
        x_local, w, out = ctx.saved_tensors
        sg, ops = ctx.sg, ctx.ops
        dy = dy.contiguous()
        g_local = ops.relu_backward(out, dy) if ctx.active else dy
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = sg.all_reduce_sum(ops.colsum(g_local))
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            g_full = sg.all_gather_rows(g_local)
            ds = ops.spmm_csr(sg.at.rowptr, sg.at.col, sg.at.val, g_full, n_cols=g_full.shape[0],
                              tag="spmm_csr_f32[bwd]")
            if ctx.needs_input_grad[1]:
                dw = sg.all_reduce_sum(ops.gemm(x_local, ds, trans_a=True))
            if ctx.needs_input_grad[0]:
                dx = ops.gemm(ds, w, trans_b=True)
        return dx, dw, db, None, None, None


def sharded_gcn_layer(x_local: torch.Tensor, weight: torch.Tensor, sg: ShardedGCNGraph,
                      bias: Optional[torch.Tensor] = None, active: bool = False, ops=None) -> torch.Tensor:
    """This rank's rows of act(A (X W) + b); gradients of W / b are all-reduced, dX stays row-sharded."""
    return _ShardedGCNLayerFn.apply(x_local, weight, bias, sg, active, ops or _hip_kernels)
