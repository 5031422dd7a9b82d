"""graph-sc on MI355X — drop-in for dance/modules/single_modality/clustering/graphsc.py:34-484
(``GraphSC`` :34-271, ``GCNAE`` :274-383, ``InnerProductDecoder`` :386-411, ``WeightedGraphConv`` :414-484).

``WeightedGraphConv`` is DGL's GraphConv(norm="both") with edge weights: the out-degree^-1/2 source scaling, the
weighted sum/mean aggregation, the in-degree^-1/2 scaling, bias and ReLU are ONE fused CSR SpMM launch
(dh_spmm_csr_f32 with colscale / rowscale / bias / act) after the MFMA GEMM; degrees are those of the block, as in
the reference (:444-449,:467-476).
"""
import os
from typing import Any, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import kernels
from ....capture import kernel_clone
from ....autograd import HipLinear, gcn_layer, linear
from ....cellgraph import DataLoader, MultiLayerFullNeighborSampler
from ....graph import CSRGraph
from ....transforms import Compose, SetConfig
from ....transforms.graph import PCACellFeatureGraph
from ...base import BaseClusteringMethod


class WeightedGraphConv(nn.Module):
    """Adaptation of the dgl GraphConv model to use edge weights (parameters ``weight`` [in, out], ``bias`` [out])."""

    def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None, allow_zero_in_degree=False):
        super().__init__()
        if norm not in ("none", "both", "right", "left"):
            raise ValueError(f'Invalid norm value. Must be either "none", "both", "right" or "left". But got "{norm}".')
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        self.weight = nn.Parameter(torch.empty(in_feats, out_feats)) if weight else None
        self.bias = nn.Parameter(torch.empty(out_feats)) if bias else None
        self.reset_parameters()
        self._activation = activation

    def reset_parameters(self):
        if self.weight is not None:
            nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def forward(self, graph, feat, weight=None, agg="sum"):
        if not self._allow_zero_in_degree and graph.has_zero_in_degree():  # answered per GRAPH when it has none (no sync per block)
            raise RuntimeError("There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                               "Adding self-loop on the input graph will resolve the issue.")
        if weight is not None and self.weight is not None:
            raise RuntimeError("External weight is provided while at the same time the module has defined its own "
                               "weight parameter. Please create the module with flag weight=False.")
        weight = self.weight if weight is None else weight
        # degree scalings of this block: computed once per block (graph-sc runs two forwards per batch, graphsc.py:202,215)
        cache = graph.__dict__.setdefault("_wgc_scales", {})
        n_dst = graph.number_of_dst_nodes()
        pad = 1 if getattr(graph, "pad_row", False) else 0  # StaticCellBlock: one padding row behind the destinations
        g = graph.__dict__.get("_csr")  # one CSRGraph (hence one transpose for the backward) per block, shared by both forwards of a batch
        if g is None:
            g = graph.__dict__["_csr"] = CSRGraph(graph.rowptr, graph.col, graph.val, n_dst + pad, graph.number_of_src_nodes())
            if pad:
                g.t_rows = n_dst  # the transpose (backward gather) is built from the real rows; the padding row holds zeros only
        if self._norm not in cache:
            colscale = rowscale = None
            if self._norm != "none":
                # "both": D_out^-1/2 on the source side (:444-449), D_in^-1/2 on the destination side (:467-471); "right" AND "left": the
                # reference only scales the source side for "both" and divides by the in-degree for every other norm (:467-474)
                rowscale, colscale = kernels.degree_scales(graph.rowptr, graph.col, n_dst, graph.number_of_src_nodes(),
                                                           kernels.DEGREE_BOTH if self._norm == "both" else kernels.DEGREE_MEAN,
                                                           n_pad=pad)  # the padding row behind the destinations scales by 1
            cache[self._norm] = (colscale, rowscale)
        colscale, rowscale = cache[self._norm]
        relu = self._activation in (F.relu, torch.relu) or isinstance(self._activation, nn.ReLU)
        rst = gcn_layer(feat, weight, g, self.bias, relu, rowscale=rowscale, colscale=colscale,
                        reduce=kernels.REDUCE_MEAN if agg == "mean" else kernels.REDUCE_SUM)
        if pad:
            rst = rst[:n_dst]
        if self._activation is not None and not relu:
            rst = self._activation(rst)
        return rst


class WeightedGraphConvAlpha(WeightedGraphConv):
    """graphsc.py:487-566: the GraphConv variant whose messages are scaled by a learnable ``alpha`` per edge TYPE instead of the edge
    weight — ``alpha[gene id]`` for gene -> cell and cell -> gene edges, ``alpha[gene_num]`` between two genes, ``alpha[gene_num + 1]``
    otherwise (self loops of cells; the rule of scDeepSort's AdaptiveSAGE), node ids in ``srcdata["id"]`` / ``dstdata["id"]`` (genes >= 0,
    cells < 0).  Defined but not instantiated by the reference's GCNAE; evaluated here as one SpMM with per-edge values (and an SDDMM
    for alpha's gradient) instead of an [E, D] message tensor.  As in the reference, the module's own ``weight`` is never applied — only
    an external one is (:534-542)."""

    def forward(self, graph, feat, weight=None, alpha=None, gene_num=None):
        from ....autograd import edge_weighted_sum, linear
        if not self._allow_zero_in_degree and graph.has_zero_in_degree():
            raise RuntimeError("There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                               "Adding self-loop on the input graph will resolve the issue.")
        feat_src = feat[0] if isinstance(feat, tuple) else feat
        n_dst, n_src = graph.number_of_dst_nodes(), graph.number_of_src_nodes()
        rowscale = colscale = None
        if self._norm != "none":
            rowscale, colscale = kernels.degree_scales(graph.rowptr, graph.col, n_dst, n_src,
                                                       kernels.DEGREE_BOTH if self._norm == "both" else kernels.DEGREE_MEAN)
        if colscale is not None:
            feat_src = feat_src * colscale[:, None]
        if weight is not None:
            if self.weight is not None:
                raise RuntimeError("External weight is provided while at the same time the module has defined its own weight parameter. "
                                   "Please create the module with flag weight=False.")
            feat_src = linear(feat_src, weight.t())
        src_id, dst_id = graph.srcdata["id"].reshape(-1).to(torch.int64), graph.dstdata["id"].reshape(-1).to(torch.int64)  # [N] or [N, 1]
        rows = torch.repeat_interleave(torch.arange(n_dst, device=graph.col.device), (graph.rowptr[1:n_dst + 1] - graph.rowptr[:n_dst]).to(torch.int64),
                                       output_size=graph.col.numel())
        sid, did = src_id[graph.col.to(torch.int64)], dst_id[rows]
        idx = torch.full_like(sid, gene_num + 1)
        idx = torch.where((sid >= 0) & (did < 0), sid, idx)       # gene -> cell
        idx = torch.where((did >= 0) & (sid < 0), did, idx)       # cell -> gene
        idx = torch.where((did >= 0) & (sid >= 0), torch.full_like(idx, gene_num), idx)  # gene - gene
        g = CSRGraph(graph.rowptr, graph.col, None, n_dst, n_src)
        rst = edge_weighted_sum(feat_src, alpha.reshape(-1)[idx], g)
        if rowscale is not None:
            rst = rst * rowscale[:, None]
        if self.bias is not None:
            rst = rst + self.bias
        if self._activation is not None:
            rst = self._activation(rst)
        return rst


class InnerProductDecoder(nn.Module):

    def __init__(self, activation=torch.sigmoid, dropout=0.1):
        super().__init__()
        self.dropout = dropout
        self.activation = activation

    def forward(self, z):
        z = F.dropout(z, self.dropout)  # training=True always, as in the reference (:409)
        return self.activation(linear(z, z))  # z z^T on the matrix cores


class GCNAE(nn.Module):

    def __init__(self, *, agg: str, activation: str, in_feats: int, n_hidden: int, hidden_dim: int, hidden_1: int,
                 hidden_2: int, dropout: float, n_layers: int, hidden_relu: bool, hidden_bn: bool):
        super().__init__()
        self.agg = agg
        activation = {"gelu": F.gelu, "prelu": F.prelu, "relu": F.relu, "leaky_relu": F.leaky_relu}.get(activation, activation)
        hidden = None if n_hidden == 0 else [hidden_1] if n_hidden == 1 else [hidden_1, hidden_2]
        self.dropout = nn.Dropout(p=dropout) if dropout != 0 else None
        self.layer1 = WeightedGraphConv(in_feats=in_feats, out_feats=hidden_dim, activation=activation)
        if n_layers == 2:
            self.layer2 = WeightedGraphConv(in_feats=hidden_dim, out_feats=hidden_dim, activation=activation)
        self.decoder = InnerProductDecoder(activation=lambda x: x)
        self.decoder.linear_logits = True  # adj_logits = z z^T exactly: GraphSC.fit may use the fused decoder loss
        self.embedding_dim = hidden_dim if hidden is None else hidden[-1]
        self.hidden = hidden
        if hidden is not None:
            enc = []
            for i, s in enumerate(hidden):
                enc.append(HipLinear(hidden_dim if i == 0 else hidden[i - 1], hidden[i]))
                if hidden_bn and i != len(hidden):
                    enc.append(nn.BatchNorm1d(hidden[i]))
                if hidden_relu and i != len(hidden):
                    enc.append(nn.ReLU())
            self.encoder = nn.Sequential(*enc)

    def forward(self, blocks, features, decode: bool = True):
        """``decode=False`` returns (None, embedding): GraphSC.fit's first forward of a batch only keeps the embedding
        (graphsc.py:202-203 discards ``adj_logits``), so the B x B product need not be formed there."""
        x = blocks[0].srcdata["features"]
        for i in range(len(blocks)):
            if self.dropout is not None:
                x = self.dropout(x)
            x = (self.layer1 if i == 0 else self.layer2)(blocks[i], x, agg=self.agg)
        if self.hidden is not None:
            x = self.encoder(x)
        return (self.decoder(x) if decode else None), x

    def forward_sharded(self, scg, features):
        """The embedding of this rank's cells from a pass over the WHOLE graph with the cells sharded by range and the genes replicated
        (``sharding.ShardedCellGeneGraph``) — the blocks the full-neighbour sampler builds for the seed set "all cells":
        every node is a destination of the inner layers (their gene rows: partial sums over the rank's cells, completed by an all-reduce
        of G x D floats), the cells of the last.  ``features``: rows of the local nodes (genes, then this rank's cells).  Dropout
        draws the replicated gene rows from a generator every rank seeds alike (``scg.gene_rng``), the cell rows from the rank's own."""
        from ....sharding import sharded_batch_norm, sharded_cellgene_conv
        x = features
        g = scg.n_genes
        n_l = 1 + (1 if hasattr(self, "layer2") else 0)
        for i in range(n_l):
            if self.dropout is not None and self.training and self.dropout.p > 0:
                keep = 1.0 - self.dropout.p
                mg = (torch.rand((g, x.shape[1]), device=x.device, generator=scg.gene_rng) < keep)
                mc = (torch.rand((x.shape[0] - g, x.shape[1]), device=x.device) < keep)
                x = x * torch.cat((mg, mc)).to(x.dtype) / keep
            layer = self.layer1 if i == 0 else self.layer2
            relu = layer._activation in (F.relu, torch.relu) or isinstance(layer._activation, nn.ReLU)
            x = sharded_cellgene_conv(x, layer.weight, layer.bias, scg, "all" if i < n_l - 1 else "cells", norm=layer._norm, agg=self.agg, relu=relu)
            if layer._activation is not None and not relu:
                x = layer._activation(x)
        if self.hidden is not None:
            for m in self.encoder:  # the rows are cells now: BatchNorm statistics over ALL cells
                x = sharded_batch_norm(m, x, scg.n_cells, scg.group) if isinstance(m, nn.BatchNorm1d) else m(x)
        return x


def block_dst_adjacency(block) -> torch.Tensor:
    """``g.adjacency_matrix().to_dense()[dst][:, dst]`` of the reference (:208-209): B x B, entry [u, v] = 1 for an
    edge u -> v between two destination nodes of the block."""
    b = block.number_of_dst_nodes()
    cols = block.col.to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(b, device=block.rowptr.device), block.in_degrees(), output_size=cols.numel())
    keep = cols < b  # edges whose source is itself a destination node; the others add 0 at a clamped position (no host sync)
    adj = torch.zeros((b, b), dtype=torch.float32, device=block.rowptr.device)
    adj.index_put_((cols.clamp(max=b - 1), rows), keep.to(torch.float32), accumulate=True)
    return adj


def block_dst_edges(block):
    """The edges of ``block_dst_adjacency`` as lists, without a host round trip: (u, v, m) over ALL block edges, m = 1 where
    the source is itself a destination node of the block (u < B) and 0 otherwise (u is then clamped into range)."""
    b = block.number_of_dst_nodes()
    cols = block.col.to(torch.int64)
    rows = torch.repeat_interleave(torch.arange(b, device=block.rowptr.device), block.in_degrees(), output_size=cols.numel())
    return cols.clamp(max=b - 1), rows, (cols < b).to(torch.float32)


class _SparseTargetBCE(torch.autograd.Function):
    """mean(binary_cross_entropy_with_logits(x, y, pos_weight=p)) for a target y that is m[e] at (u[e], v[e]) and 0 elsewhere
    (entries listed at most once) — graphsc.py:214-216 without the dense B x B target and torch's ~15 elementwise passes over
    the logits: element loss softplus(x) (y = 0) resp. p * softplus(-x) (y = 1), evaluated as one fused pass over x
    (dh_softplus_rowsum_f32) plus a correction on the edge list; backward = one pass (dh_sigmoid_scale_f32) + the same
    correction."""

    @staticmethod
    def forward(ctx, x, u, v, m, p):
        xe = x[u, v]
        dense = kernels.softplus_rowsum(x).sum(dtype=torch.float64)
        corr = (m * (p * F.softplus(-xe) - F.softplus(xe))).sum(dtype=torch.float64)
        ctx.save_for_backward(x, u, v, m, p, xe)
        return ((dense + corr) / x.numel()).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        x, u, v, m, p, xe = ctx.saved_tensors
        scale = (g / x.numel()).to(torch.float32)
        dx = kernels.sigmoid_scale(x, scale)
        sig = torch.sigmoid(xe)
        dx.index_put_((u, v), m * (p * (sig - 1) - sig) * scale, accumulate=True)
        return dx, None, None, None, None


def sparse_target_bce(logits, u, v, m, pos_weight):
    return _SparseTargetBCE.apply(logits.contiguous(), u, v, m, pos_weight)


class _GramListedBCE(torch.autograd.Function):
    """mean(binary_cross_entropy_with_logits(z z^T, y, pos_weight=p)) for a target y that is 1 at the listed (us[e], vs[e])
    (each at most once) and 0 elsewhere, as a function of z and without the B x B logits: the dense part (sum of
    softplus(<z_i, z_j>) and its gradient 2 sum_j sigmoid(<z_i, z_j>) z_j) is ONE matrix-core kernel that keeps every logit
    tile in registers (dh_gram_sigmoid_f32); the y = 1 corrections need the logits of the listed entries only, recomputed as dot
    products by dh_gram_listed_forward_f32 / _backward_f32 (two launches; fixed accumulation order).  Replaces the z z^T GEMM,
    the two passes over the logits and the two B x B x d GEMMs of the backward of graphsc.py:208-216 / :405-411.
    ``us`` / ``vs`` int32, ``p`` a python float."""

    @staticmethod
    def forward(ctx, z, us, vs, p, diag=False):
        n = z.shape[0]
        rowloss, o = kernels.gram_sigmoid(z)
        xe, term = kernels.gram_listed_forward(z, us, vs, p)
        ctx.save_for_backward(z, o, us, vs, xe)
        ctx.p = p
        ctx.diag = bool(diag) and us.numel() == n  # the caller vouches for us = vs = arange(n): the elementwise backward
        return ((rowloss.sum(dtype=torch.float64) + term.sum(dtype=torch.float64)) / float(n * n)).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        z, o, us, vs, xe = ctx.saved_tensors
        scale = (g / float(z.shape[0]**2)).to(torch.float32)
        if ctx.diag:
            return kernels.gram_diag_backward(z, o, xe, ctx.p, scale), None, None, None, None
        return kernels.gram_listed_backward(z, o, us, vs, xe, ctx.p, scale), None, None, None, None


def gram_listed_bce(z, us, vs, pos_weight, diag: bool = False):
    """``diag=True``: the listed entries are exactly (i, i) for every row i (us = vs = arange(n)) — same value and gradient, the backward's
    correction is then an elementwise pass instead of a scan of the list per row."""
    return _GramListedBCE.apply(z.contiguous(), us.to(torch.int32), vs.to(torch.int32), float(pos_weight), diag)


class _AggFirstConv(torch.autograd.Function):
    """relu((A_norm X) W + b) for an aggregated, gradient-free input ``ax`` = A_norm X (dh_graphsc_steps phase 3): the WeightedGraphConv of a
    batch in the aggregate-first order — one product on the matrix cores, and a backward that is dW = ax^T (dy o [y > 0]), db = column sums:
    the transposed block graphsc.py's multiply-first order needs for dX W^T never exists (the input features carry no gradient)."""

    @staticmethod
    def forward(ctx, ax, weight, bias):
        y = kernels.gemm(ax, weight, bias=bias.detach(), act=kernels.ACT_RELU)
        ctx.save_for_backward(ax, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        ax, y = ctx.saved_tensors
        g = kernels.relu_backward(y, dy.contiguous())
        return None, kernels.gemm(ax, g, trans_a=True), kernels.colsum(g)


class _PinnedCounts:
    """A small ring of pinned int64 scalars for asynchronous count reads: a slot is handed out again ``slots`` batches later,
    long after its value was read (the loader runs one batch ahead)."""

    def __init__(self, slots: int = 4):
        self._slots, self._n, self._next = [], slots, 0

    def take(self) -> torch.Tensor:
        if not self._slots:
            self._slots = [torch.empty(1, dtype=torch.int64).pin_memory() for _ in range(self._n)]
        slot = self._slots[self._next]
        self._next = (self._next + 1) % self._n
        return slot


_PINNED_COUNTS = _PinnedCounts()


def _dst_edge_hook(blocks):
    """DataLoader block hook of GraphSC.fit: the edges of the last block whose source is itself a destination node — the
    non-zeros of ``g.adjacency_matrix().to_dense()[dst][:, dst]`` (graphsc.py:208-209) — as (edge ids with those edges
    first and in edge order, their count in pinned host memory, event).  Runs on the stream that built the block (the
    loader's side stream, one batch ahead), so reading the count later does not wait for the model's kernels."""
    last = blocks[-1]
    outside = (last.col >= last.number_of_dst_nodes()).to(torch.uint8)
    order = torch.sort(outside, stable=True).indices
    inside = (outside.numel() - outside.sum(dtype=torch.int64)).reshape(1)
    if not inside.is_cuda:  # host tensors (tests): nothing to wait for
        return order, inside, None
    cnt = _PINNED_COUNTS.take()
    cnt.copy_(inside, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return order, cnt, ev


# "fused" (default): GraphSC.fit evaluates the decoder loss by dh_gram_sigmoid_f32 + dh_gram_listed_* (no B x B logits); the listed
# target entries come from the loader's block hook (no host round trip on the model's stream).  "dense": the reference-shaped
# path (z z^T GEMM + passes over the logits), also taken when the fused kernel does not support the shape.
DECODER_MODE = os.environ.get("DANCE_AMD_GRAPHSC_DECODER", "fused")
if DECODER_MODE not in ("fused", "dense"):
    raise ValueError(f"DANCE_AMD_GRAPHSC_DECODER must be 'fused' or 'dense', got {DECODER_MODE!r}")


# One hipGraph per training step (GraphSC.fit, single process, fused decoder, one layer, a CellFeatureGraph-layout graph whose seeds
# are all cells): the static-shape block (cellgraph.StaticCellBlock), both forwards, the decoder loss, the backward and Adam are
# captured once and replayed per batch — at the reference's batch size (128) a step is ~170 launches of microsecond kernels, i.e.
# host-bound (1.36 ms / batch eager, profiles/r03e_ref_batch_epochs.json).  DANCE_AMD_HIPGRAPH=0 keeps the eager loop; the last,
# short batch of an epoch always runs eagerly.
HIPGRAPH = os.environ.get("DANCE_AMD_HIPGRAPH", "1") != "0"
HIPGRAPH_MIN_BATCHES = int(os.environ.get("DANCE_AMD_HIPGRAPH_MIN_BATCHES", "64"))  # capturing costs two eager steps + the instantiation (tens of ms): it must be amortised
HIPGRAPH_MAX_BATCH = int(os.environ.get("DANCE_AMD_HIPGRAPH_MAX_BATCH", "2048"))   # above this a step is kernel-bound and replaying it gains little

# The persistent step (dance_amd/ministep.py, csrc/ministep.hip): every full batch of an epoch behind ONE C call, four launches per step,
# straight off the graph's CSR rows (no block, no transposed copy, Adam in the gradient kernel's epilogue).  The default wherever it
# applies — the reference's own model shape (one WeightedGraphConv norm="both" + ReLU, one Linear, linear decoder) on a CellFeatureGraph-
# layout graph with cell seeds, batches up to MINISTEP_MAX_BATCH (a seed's row of z z^T is one workgroup's job there: it stops paying
# once the all-pairs decoder wants the matrix cores).  DANCE_AMD_MINISTEP=0: the captured hipGraph step / the eager loop below.
MINISTEP = os.environ.get("DANCE_AMD_MINISTEP", "1") != "0"
MINISTEP_MAX_BATCH = int(os.environ.get("DANCE_AMD_MINISTEP_MAX_BATCH", "512"))


class _CapturedStep:
    """The training step of ``GraphSC.fit`` on a ``StaticCellBlock``, captured as one ``torch.cuda.CUDAGraph`` (= hipGraph)."""

    def __init__(self, fit_self, g, batch_size: int, optim):
        from ....cellgraph import StaticCellBlock
        self.model, self.optim = fit_self.model, optim
        self.block = StaticCellBlock(g, batch_size)
        b = float(batch_size)
        # the only edges among a batch's own cells are their self loops: adj = I, adj.sum() = B (graphsc.py:208-214)
        self.pos_weight = (b * b - b) / b
        self.norm = b * b / ((b * b - b) * 2) if batch_size > 1 else 1.0
        self.diag = torch.arange(batch_size, dtype=torch.int32, device=g.device)
        self.graph = None
        self.emb = self.loss = None

    def _forward_backward(self):
        blk = self.block.rebuild()
        x = blk.srcdata["features"]
        _, emb = self.model.forward([blk], x, decode=False)  # :202
        emb_out = kernel_clone(emb.detach())  # a kernel, not a memcpy node (capture.py)
        _, emb2 = self.model.forward([blk], x, decode=False)  # :215, fresh dropout
        loss = self.norm * gram_listed_bce(F.dropout(emb2, self.model.decoder.dropout), self.diag, self.diag, self.pos_weight, diag=True)
        self.optim.zero_grad(set_to_none=True)
        loss.backward()
        return emb_out, loss.detach()

    def _optimiser_step(self):
        if not kernels.adam_step(self.optim):  # dh_adam_step_f32 once the optimiser's state exists (created by the first torch step)
            self.optim.step()

    def _step(self):
        out = self._forward_backward()
        self._optimiser_step()
        return out

    def capture(self, first_seeds: torch.Tensor, *, split: bool = False):
        """Warm up on a side stream (allocator, Adam state), then record (dance_amd/capture.py).  ``split``: two graphs with the
        gradient all-reduce between them — the data-parallel form.  Parameters and optimiser state are restored afterwards, so
        capturing leaves the model exactly where it was."""
        import copy

        from .... import sharding
        from ....capture import CapturedStep
        if self.optim.state:
            raise RuntimeError("capture expects a fresh optimiser (its state tensors are created by the warm-up steps and reset below)")
        saved_model = copy.deepcopy(self.model.state_dict())
        self.block.seeds.copy_(first_seeds)
        self.graph = CapturedStep(self._forward_backward, self._optimiser_step, first_seeds.device, split=split,
                                  between=lambda: sharding.allreduce_gradients(self.model),
                                  keep_alive=lambda: [p.grad for p in self.model.parameters() if p.grad is not None], params=list(self.model.parameters()))
        self.emb, self.loss = self.graph.outputs
        self.model.load_state_dict(saved_model)  # copies INTO the captured parameter / buffer tensors
        for st in self.optim.state.values():      # the graph updates these very tensors: reset them in place (moments 0, step 0)
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        self.block.bad.zero_()

    def run(self, seeds: torch.Tensor):
        self.block.seeds.copy_(seeds)
        self.graph.replay()
        return self.emb, self.loss


class GraphSC(BaseClusteringMethod):

    # Seed order of the mini-batches: None = shuffled on the device; a (host) torch.Generator makes the order
    # reproducible (``torch.randperm(n, generator=...)`` per epoch).  Not a constructor argument: the constructor is
    # the reference's (graphsc.py:70-87).
    shuffle_generator = None
    # Captured steps as two graphs with the gradient all-reduce between them (the form used with more than one process); True forces
    # it on one process too (tests)
    capture_split = False
    # measurement aid (scripts/bench_configs.py): True -> ``epoch_ms`` holds the device time between the epoch boundaries of the last fit
    # (the last epoch also carries the embedding's gather and copy to the host)
    record_epoch_times = False

    def __init__(self, agg: str = "sum", activation: str = "relu", in_feats: int = 50, n_hidden: int = 1, hidden_dim: int = 200,
                 hidden_1: int = 300, hidden_2: int = 0, dropout: float = 0.1, n_layers: int = 1, hidden_relu: bool = False,
                 hidden_bn: bool = False, n_clusters: int = 10, cluster_method: str = "kmeans", num_workers: int = 1,
                 device: str = "auto"):
        super().__init__()
        self.n_layers = n_layers
        self.n_clusters = n_clusters
        self.cluster_method = cluster_method
        self.num_workers = num_workers
        self.device = "cuda" if device == "auto" else device
        self.model = GCNAE(agg=agg, activation=activation, in_feats=in_feats, n_hidden=n_hidden, hidden_dim=hidden_dim,
                           hidden_1=hidden_1, hidden_2=hidden_2, dropout=dropout, n_layers=n_layers, hidden_relu=hidden_relu,
                           hidden_bn=hidden_bn).to(self.device)

    @staticmethod
    def preprocessing_pipeline(n_top_genes: int = 3000, normalize_weights: str = "log_per_cell", n_components: int = 50,
                               normalize_edges: bool = False, log_level="INFO"):
        """graphsc.py:110-146 with every step on the device (DeviceArray slots): filter genes / cells, normalize_total, log1p,
        cell_ranger HVG (n_top_genes), the optional second log1p + per-cell normalisation of the edge weights, then the PCA
        cell-gene graph.  ``PCACellFeatureGraph.pca_device`` decides where the gene PCA runs (host sklearn by default)."""
        from ....transforms import FilterCellsScanpy, FilterGenesScanpy, HighlyVariableGenesLogarithmizedByTopGenes, Log1P, NormalizeTotal
        transforms = [
            FilterGenesScanpy(min_counts=3),
            FilterCellsScanpy(min_counts=1),
            NormalizeTotal(max_fraction=1.0),
            Log1P(),
            HighlyVariableGenesLogarithmizedByTopGenes(n_top_genes=n_top_genes, flavor="cell_ranger", subset=True),
        ]
        if normalize_weights == "log_per_cell":
            transforms.extend([Log1P(), NormalizeTotal(target_sum=1, max_fraction=1.0)])
        elif normalize_weights == "per_cell":
            transforms.append(NormalizeTotal(target_sum=1, max_fraction=1.0))
        elif normalize_weights != "none":
            raise ValueError(f"Unknown normalization option {normalize_weights!r}."
                             "Available options are: 'none', 'log_per_cell', 'per_cell'")
        transforms.extend([
            PCACellFeatureGraph(n_components=n_components, normalize_edges=normalize_edges, feat_norm_mode="standardize"),
            SetConfig({"feature_channel": "CellFeatureGraph", "feature_channel_type": "uns", "label_channel": "Group"}),
        ])
        return Compose(*transforms, log_level=log_level)

    def fit(self, g, y: Optional[Any] = None, *, epochs: int = 100, lr: float = 1e-5, batch_size: int = 128,
            show_epoch_ari: bool = False, eval_epoch: bool = False):
        with kernels.mini_batch_products():  # the step's small fp32 products on dh_gemm_f32_small (a captured step records them so)
            return self._fit(g, y, epochs=epochs, lr=lr, batch_size=batch_size, show_epoch_ari=show_epoch_ari, eval_epoch=eval_epoch)

    def _fit(self, g, y, *, epochs, lr, batch_size, show_epoch_ari, eval_epoch):
        g = g.to(self.device)
        g.ndata["order"] = g.ndata["label"] = g.ndata["feat_id"]
        train_ids = np.where(g.ndata["label"].cpu().numpy() != -1)[0]
        # more than one process (torch.distributed initialised, one rank per GPU): plain data parallelism over the seed cells
        # (BASELINE config 4) — every rank samples blocks for its share of the cells from the replicated graph, gradients are
        # averaged with one flat all-reduce per step, embeddings are gathered at the end of an epoch
        from .... import sharding
        rank, world = sharding.world_info()
        if world > 1:
            sharding.broadcast_parameters(self.model)
            train_ids = sharding.shard_seed_ids(torch.from_numpy(train_ids)).numpy()
        sampler = MultiLayerFullNeighborSampler(self.n_layers)
        if DECODER_MODE not in ("fused", "dense"):
            raise ValueError(f"unknown decoder mode {DECODER_MODE!r} ('fused' or 'dense')")
        fused = (DECODER_MODE == "fused" and getattr(self.model.decoder, "linear_logits", False)
                 and kernels.gram_sigmoid_supported(batch_size, self.model.embedding_dim))
        dataloader = DataLoader(g, train_ids, sampler, batch_size=batch_size, shuffle=True, drop_last=False,
                                generator=self.shuffle_generator, block_hook=_dst_edge_hook if fused else None)
        n_full = len(train_ids) // batch_size
        use_graph = (HIPGRAPH and fused and self.n_layers == 1 and dataloader.cells_only and g.device.type == "cuda"
                     and n_full >= HIPGRAPH_MIN_BATCHES and 1 < batch_size <= HIPGRAPH_MAX_BATCH)
        # fused: one multi-tensor kernel per step instead of ~14 — inside the captured step of batch 128 that is 0.57 -> 0.47 ms per batch
        optim = torch.optim.Adam(self.model.parameters(), lr=lr, capturable=use_graph, fused=g.device.type == "cuda")
        captured = None
        from ....ministep import GraphSCStepper
        use_mini = (MINISTEP and fused and dataloader.cells_only and g.device.type == "cuda" and n_full >= 1 and 1 < batch_size <= MINISTEP_MAX_BATCH
                    and not self.capture_split and GraphSCStepper.eligible(self.model, g, batch_size, optim))
        stepper = GraphSCStepper(self.model, g, batch_size, optim, world) if use_mini else None
        # batches past MINISTEP_MAX_BATCH keep the persistent step's FIRST half — the aggregation straight off the CSR rows, both forwards,
        # no block / degree kernels / transposed copy (dh_graphsc_steps phase 3) — and run the dense layers, the all-pairs decoder and Adam
        # on the big-tile kernels through autograd ("aggfirst"): ~40 launches per batch instead of ~150 and no host read-back
        use_agg = (MINISTEP and not use_mini and fused and dataloader.cells_only and g.device.type == "cuda" and n_full >= 1 and batch_size > MINISTEP_MAX_BATCH
                   and not self.capture_split and GraphSCStepper.eligible(self.model, g, batch_size, optim))
        if use_agg:
            stepper = GraphSCStepper(self.model, g, batch_size, optim, 1)
        use_graph = use_graph and not use_mini and not use_agg
        self.step_mode = "ministep" if use_mini else "aggfirst" if use_agg else "hipgraph" if use_graph else "eager"
        self.losses, aris, Z = [], [], {}
        marks = []
        for epoch in range(epochs):
            if self.record_epoch_times and torch.cuda.is_available():  # measurement aid: see the class attribute
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record()
            self.model.train()
            z, order, losses = [], [], []
            if use_mini:
                # the loader's own seed order (``DataLoader.__iter__``), all full batches of the epoch in one call
                idx = dataloader.indices
                perm = (torch.randperm(idx.numel(), device=idx.device) if self.shuffle_generator is None else
                        torch.randperm(idx.numel(), generator=self.shuffle_generator).to(idx.device))
                idx = idx[perm].contiguous()
                z_all = torch.empty((n_full * batch_size, self.model.embedding_dim), dtype=torch.float32, device=g.device)
                loss_all = torch.empty(n_full, dtype=torch.float32, device=g.device)
                stepper.run(idx, n_full, z_all, loss_all)
                stepper.check_flags("GraphSC.fit")  # one read per epoch
                z.append(z_all)
                order.append(g.ndata["order"][idx[:n_full * batch_size]])
                losses.extend(loss_all.unbind(0))
                batches = []
                if n_full * batch_size < idx.numel():  # the short last batch runs eagerly
                    tail = sampler.sample(g, idx[n_full * batch_size:], True)
                    tail[2][-1].hook_out = _dst_edge_hook(tail[2])
                    batches = [tail]
            elif use_agg:
                idx = dataloader.indices
                perm = (torch.randperm(idx.numel(), device=idx.device) if self.shuffle_generator is None else
                        torch.randperm(idx.numel(), generator=self.shuffle_generator).to(idx.device))
                idx = idx[perm].contiguous()
                z_all = torch.empty((n_full * batch_size, self.model.embedding_dim), dtype=torch.float32, device=g.device)
                l1, enc = self.model.layer1, self.model.encoder[0]
                b = float(batch_size)
                pos_weight, norm = (b * b - b) / b, b * b / ((b * b - b) * 2)  # graphsc.py:210-214 with adj = I (the seeds' self loops; checked per epoch)
                diag = torch.arange(batch_size, dtype=torch.int32, device=g.device)
                for i in range(n_full):
                    ax = stepper.aggregate(idx[i * batch_size:(i + 1) * batch_size])
                    with torch.no_grad():  # :202-203 — only the embedding is kept
                        h0 = kernels.gemm(ax[0], l1.weight, bias=l1.bias, act=kernels.ACT_RELU)
                        kernels.gemm(h0, enc.weight, trans_b=True, bias=enc.bias, out=z_all[i * batch_size:(i + 1) * batch_size])
                    emb2 = enc(_AggFirstConv.apply(ax[1], l1.weight, l1.bias))  # :215, fresh dropout
                    loss = norm * gram_listed_bce(F.dropout(emb2, self.model.decoder.dropout), diag, diag, pos_weight, diag=True)
                    optim.zero_grad(set_to_none=True)
                    loss.backward()
                    sharding.allreduce_gradients(self.model)
                    if not kernels.adam_step(optim):
                        optim.step()
                    losses.append(loss.detach())
                stepper.check_flags("GraphSC.fit")
                z.append(z_all)
                order.append(g.ndata["order"][idx[:n_full * batch_size]])
                batches = []
                if n_full * batch_size < idx.numel():  # the short last batch runs through the general loop
                    tail = sampler.sample(g, idx[n_full * batch_size:], True)
                    tail[2][-1].hook_out = _dst_edge_hook(tail[2])
                    batches = [tail]
            elif use_graph:
                # same seed order as the loader's (``DataLoader.__iter__``): one permutation per epoch from the same generator
                idx = dataloader.indices
                perm = (torch.randperm(idx.numel(), device=idx.device) if self.shuffle_generator is None else
                        torch.randperm(idx.numel(), generator=self.shuffle_generator).to(idx.device))
                idx = idx[perm]
                if captured is None:
                    captured = _CapturedStep(self, g, batch_size, optim)
                    captured.capture(idx[:batch_size], split=world > 1 or self.capture_split)
                z_all = torch.empty((n_full * batch_size, self.model.embedding_dim), dtype=torch.float32, device=g.device)
                loss_all = torch.empty(n_full, dtype=torch.float32, device=g.device)
                for i in range(n_full):
                    emb, loss = captured.run(idx[i * batch_size:(i + 1) * batch_size])
                    z_all[i * batch_size:(i + 1) * batch_size].copy_(emb)
                    loss_all[i].copy_(loss)
                if int(captured.block.bad) != 0:
                    raise RuntimeError("GraphSC.fit: a seed of the captured step is not a cell of a CellFeatureGraph-layout graph with one self loop "
                                       "(set DANCE_AMD_HIPGRAPH=0 for graphs with other in-neighbours)")
                z.append(z_all)
                order.append(g.ndata["order"][idx[:n_full * batch_size]])
                losses.extend(loss_all.unbind(0))
                batches = []
                if n_full * batch_size < idx.numel():  # the short last batch runs eagerly
                    tail = sampler.sample(g, idx[n_full * batch_size:], True)
                    tail[2][-1].hook_out = _dst_edge_hook(tail[2])
                    batches = [tail]
            else:
                batches = dataloader
            for input_nodes, output_nodes, blocks in batches:
                input_features = blocks[0].srcdata["features"]
                last = blocks[-1]
                _, emb = self.model.forward(blocks, input_features, decode=False)  # :202 (its adj_logits are never used)
                z.append(emb.detach())
                order.append(last.dstdata["order"])
                if fused:
                    # adj = g.adjacency_matrix().to_dense()[dst][:, dst] (:208-209) is zero except for the edges among the
                    # batch's own cells: _dst_edge_hook listed them one batch ahead on the loader's stream, so their count —
                    # adj.sum() of :210-213 — is on the host by now and pos_weight / norm are plain floats
                    order_e, cnt, ev = last.hook_out
                    if ev is not None:
                        ev.synchronize()
                    n_listed = int(cnt)
                    sel = order_e[:n_listed]
                    us = last.col[sel].to(torch.int64)
                    vs = torch.searchsorted(last.rowptr, sel.to(last.rowptr.dtype), right=True).to(torch.int64) - 1
                    total = float(last.number_of_dst_nodes())**2
                    pos_weight = (total - n_listed) / n_listed if n_listed else float("inf")
                    factor = (total - n_listed) * 2
                    norm = total / (factor if factor != 0 else 1.0)
                    # second forward, fresh dropout (:215); the decoder's own dropout (:409) is the last draw, as in the reference
                    _, emb2 = self.model.forward(blocks, input_features, decode=False)
                    loss = norm * gram_listed_bce(F.dropout(emb2, self.model.decoder.dropout), us, vs, pos_weight)
                    if not n_listed:  # the reference's 0 * inf (pos_weight = inf against an all-zero target)
                        loss = loss * float("nan")
                else:
                    # the loss scalars of graphsc.py:208-214 stay on the device (the reference reads them back every batch);
                    # adj is kept as an edge list (block_dst_edges) and never materialised (block_dst_adjacency is the dense form)
                    eu, ev_, em = block_dst_edges(last)
                    total = float(last.number_of_dst_nodes())**2
                    s_ = em.sum()
                    pos_weight = ((total - s_) / s_).reshape(1)
                    factor = (total - s_) * 2
                    norm = total / torch.where(factor == 0, torch.ones_like(factor), factor)
                    adj_logits, _ = self.model.forward(blocks, input_features)  # second forward, fresh dropout (:215)
                    loss = norm * sparse_target_bce(adj_logits, eu, ev_, em, pos_weight)
                # next to a captured step the gradients stay the tensors the graph writes (zeroed and accumulated into in place): an eager
                # step that replaced them (set_to_none) made the following replays fault at 100k cells (dance_amd/capture.py)
                optim.zero_grad(set_to_none=captured is None)
                loss.backward()
                sharding.allreduce_gradients(self.model)
                optim.step()
                losses.append(loss.detach())
            self.losses.extend(torch.stack(losses).tolist() if losses else [])
            if eval_epoch or epoch == epochs - 1:  # the embedding only leaves the device when somebody reads it
                zc, oc = sharding.gather_embeddings(torch.cat(z), torch.cat(order))  # one process: a sort by cell order
                self.z = zc.cpu().numpy()
            if eval_epoch and y is not None:
                aris.append(self.score(None, y))
                Z[f"epoch{epoch}"] = self.z
        if marks:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
            marks[-1].synchronize()
            self.epoch_ms = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]
        if eval_epoch and aris:
            self.z = Z[f"epoch{int(np.argmax(aris))}"]

    def fit_full_graph(self, g, y: Optional[Any] = None, *, epochs: int = 100, lr: float = 1e-5, eval_epoch: bool = False):
        """``fit`` with the whole cell set as ONE batch per epoch (what graphsc.py:148-246 does for ``batch_size >= n_cells``: the blocks
        of the full-neighbour sampler are then the graph itself), with the cells sharded by contiguous range over the ranks of
        ``torch.distributed`` and the genes replicated (BASELINE config 4 as stated: destination-sharded).  Per epoch and
        layer: cell rows aggregate locally, gene rows are partial sums completed by an all-reduce of G x D floats; the decoder loss over
        all n x n cell pairs is evaluated row block by row block against the all-gathered embedding (no n x n matrix, no gradient
        exchange); the replicated parameters' gradients are summed over the ranks.  One process: the same arithmetic without the
        collectives.  ``g``: the whole ``CellGeneGraph`` (genes-first layout) on every rank at set-up."""
        from .... import sharding
        g = g.to(self.device)
        g.ndata["order"] = g.ndata["label"] = g.ndata["feat_id"]
        rank, world = sharding.world_info()
        scg = sharding.ShardedCellGeneGraph.from_global(g, ops=getattr(self, "ops", None))
        if world > 1:
            sharding.broadcast_parameters(self.model)
        seed = torch.randint(0, 2**31 - 1, (1, ), device=self.device)
        if world > 1:
            import torch.distributed as dist
            dist.broadcast(seed, src=0)
        scg.gene_rng = torch.Generator(device=self.device).manual_seed(int(seed))
        loc, ng = scg.local, scg.n_genes
        feats = loc.ndata["features"]
        # the decoder's target: adj[dst][:, dst] over the cells — in a cell - gene graph the cells' self loops (graphsc.py:208-213)
        rp, col = loc.rowptr.to(torch.int64), loc.col.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(loc.number_of_nodes(), device=rp.device), rp[1:] - rp[:-1])
        cell_edges = (rows >= ng) & (col >= ng)
        if bool((cell_edges & (rows != col)).any()):
            raise NotImplementedError("fit_full_graph: edges between different cells (the sharded decoder target lists self loops only)")
        has_self = torch.zeros(loc.number_of_nodes(), dtype=torch.bool, device=rp.device)
        has_self[rows[cell_edges]] = True
        has_self = has_self[ng:]
        n_listed = int(scg.all_sum(has_self.sum().to(torch.float32).reshape(1)))
        total = float(scg.n_cells)**2
        pos_weight = (total - n_listed) / n_listed if n_listed else float("inf")
        factor = (total - n_listed) * 2
        norm = total / (factor if factor != 0 else 1.0)
        optim = torch.optim.Adam(self.model.parameters(), lr=lr)
        params = [p for p in self.model.parameters() if p.requires_grad]
        order = loc.ndata["order"][ng:]
        self.losses, aris, Z = [], [], {}
        for epoch in range(epochs):
            self.model.train()
            with torch.no_grad():  # :202 — kept only as z_epoch: no autograd graph over the whole cell set for it (BatchNorm's running
                emb = self.model.forward_sharded(scg, feats)  # statistics and the dropout draws are those of a grad-mode forward)
            z_epoch = emb.detach()
            emb2 = self.model.forward_sharded(scg, feats)  # second forward, fresh dropout (:215)
            loss = norm * sharding.sharded_selfloop_gram_bce(F.dropout(emb2, self.model.decoder.dropout), has_self, pos_weight, scg)
            if not n_listed:
                loss = loss * float("nan")
            optim.zero_grad()
            loss.backward()
            sharding.allreduce_sum_gradients(params)
            optim.step()
            self.losses.append(float(scg.all_sum(loss.detach().reshape(1))))
            if eval_epoch or epoch == epochs - 1:
                z_all, o_all = scg.all_gather_cells(z_epoch), scg.all_gather_cells(order.reshape(-1, 1).to(torch.float32))
                self.z = z_all[torch.argsort(o_all.reshape(-1))].cpu().numpy()
            if eval_epoch and y is not None:
                aris.append(self.score(None, y))
                Z[f"epoch{epoch}"] = self.z
        if eval_epoch and aris:
            self.z = Z[f"epoch{int(np.argmax(aris))}"]

    def predict(self, x: Optional[Any] = None):
        if self.cluster_method == "kmeans":
            from sklearn.cluster import KMeans
            return KMeans(n_clusters=self.n_clusters, init="k-means++", random_state=5, n_init=10).fit_predict(self.z)
        if self.cluster_method == "leiden":
            return run_leiden(self.z, device=self.device)
        raise ValueError(f"Unknown clustering {self.cluster_method}, available options are: 'kmeans', 'leiden'")

    def get_latent(self):
        return self.z


def run_leiden(data, device="cuda"):
    """graphsc.py:568-587: ``sc.pp.neighbors(use_rep="X", n_neighbors=300, n_pcs=0)`` + ``sc.tl.leiden``.  Neighbour
    graph on the GPU; the Leiden algorithm itself on the host (dance_amd/utils/community.py; seeded-stochastic like leidenalg, not pinned to it)."""
    from ....utils.community import leiden_like
    return [int(x) for x in leiden_like(np.asarray(data, dtype=np.float32), 300, resolution=1.0, device=device)]
