"""Autograd glue for the GCN layer op  out = act(A @ (X @ W) + b)  (SURVEY.md §0.3, §3.3).

Forward and backward are sequences of libdancehip kernels on the current HIP stream:

    forward : support = X W                 dh_gemm_f32            (MFMA f32)
              out = act(A support + b)      dh_spmm_csr_f32        (HBM-bound gather, fused epilogue)
    backward: G  = dY * (out > 0)           dh_relu_backward_f32   (only when act = relu)
              dS = A^T G                    dh_spmm_csr_f32 on the cached transposed CSR
              dW = X^T dS                   dh_gemm_f32 (trans_a, split-K)
              dX = dS W^T                   dh_gemm_f32 (trans_b)  (only if X needs grad)
              db = colsum(G)                dh_colsum_f32          (only with bias)

This is the arithmetic torch autograd performs for the reference layers
(dance/modules/single_modality/clustering/scdsc.py:496-501, dance/modules/spatial/spatial_domain/spagcn.py:357-363).
"""
import os
from typing import Optional

import torch

from . import kernels
from .graph import CSRGraph


# ReLU mask in the backward SpMM of the fused layer: "premask" (default) = dy masked once into a scratch matrix by the streaming
# dh_relu_mask_apply_f32, then the plain SpMM — 4 requests per gathered neighbour instead of 5; measured 0.70 + 4.53 = 5.23 ms
# against 5.58 ms at the headline shape although it moves 4 GB more (profiles/r03a_bwd_mask_ab.json), bit-identical;
# "fused" = the mask applied to every gathered row inside dh_spmm_csr_relu_f32 (what the sharded halo path still does,
# where the masked rows go straight into the send buffer).
BWD_MASK_MODE = os.environ.get("DANCE_AMD_BWD_MASK", "premask")


# Narrow layers (in < 64, out <= 64: SpaGCN's 50 -> 50 GraphConvolution, BASELINE config 5) run aggregate-first on two fused
# kernels (csrc/gcn_narrow.hip) from NARROW_MIN_ROWS rows on; DANCE_AMD_NARROW_FUSED=0 keeps the generic transform-first chain.
NARROW_FUSED = os.environ.get("DANCE_AMD_NARROW_FUSED", "1") != "0"
NARROW_MIN_ROWS = 4096


# Software pipeline of the wide fused layer (relu(A (X W)), width % 128 == 0).  The GEMM is bound by the matrix cores and the
# aggregation by HBM, and they saturate different units of a CU — but run one after the other they add up (the serial floor
# of the headline: 26.0 ms of MFMA time + 8.2 ms of HBM time).  The aggregation is column-sliced anyway (128-column passes,
# spmm.hip), and slice c of A S only needs the columns c of S = X W: so the GEMM is issued per column slice on the caller's
# stream in the 128 x 128 macro-tile configuration (two blocks per CU at ~170 VGPRs, which leaves a third of every SIMD's
# registers and 16 wave slots per CU free), and the aggregation of slice c runs on a second stream next to the GEMM of
# slice c + 1; backward the same way round (A^T slice c, then dW[:, c] = X^T dS[:, c] next to the gather of slice c + 1).
# Arithmetic, summation order and therefore every output bit are those of the unpipelined layer.
#   DANCE_AMD_LAYER_PIPELINE = "off" | "auto" | comma-separated slice widths (multiples of 128 that sum to the layer width)
PIPELINE = os.environ.get("DANCE_AMD_LAYER_PIPELINE", "off")
PIPELINE_MIN_ROWS = 1 << 17  # below this a layer is a handful of waves of tiles: nothing to overlap
PIPELINE_TILE = kernels.GEMM_TILE_128
PIPELINE_SIDE_PRIORITY = 0   # torch stream priority of the aggregation stream (0 = default, -1 = high)
# (workgroups, shape) of the resident aggregation kernel that runs NEXT TO a GEMM (kernels.spmm_csr_relu); None = the one-shot
# grid everywhere (round 3's pipeline, which measured serial time: its small workgroups displace the GEMM's, see spmm.hip).
# The slice that has no GEMM beside it (the last one forward, the first one backward) always runs as the one-shot grid.
PIPELINE_RESIDENT = None
_SIDE_STREAMS = {}


def _side_stream(device: torch.device) -> torch.cuda.Stream:
    key = (device.index if device.index is not None else torch.cuda.current_device(), PIPELINE_SIDE_PRIORITY)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=PIPELINE_SIDE_PRIORITY)
    return _SIDE_STREAMS[key]


def pipeline_slices(n_rows: int, width: int):
    """Column-slice boundaries [(c0, c1), ...] of the pipelined layer, or None when the layer runs unpipelined."""
    if PIPELINE == "off" or width % 128 != 0 or width < 256:
        return None
    if PIPELINE == "auto":
        if n_rows < PIPELINE_MIN_ROWS:
            return None
        widths = [128] * (width // 128)
    else:
        widths = [int(t) for t in PIPELINE.split(",")]
        if any(w <= 0 or w % 128 for w in widths) or sum(widths) != width:
            raise ValueError(f"DANCE_AMD_LAYER_PIPELINE={PIPELINE!r}: slice widths must be multiples of 128 summing to {width}")
    if len(widths) < 2:
        return None
    out, c = [], 0
    for w in widths:
        out.append((c, c + w))
        c += w
    return out


def _pipelined_forward(x, w, graph, mask, slices):
    """S = X W and Y = relu(A S) (+ sign mask), GEMM of slice c + 1 on the current stream next to the aggregation of slice c."""
    main, side = torch.cuda.current_stream(x.device), _side_stream(x.device)
    n, h = x.shape[0], w.shape[1]
    support = torch.empty((n, h), dtype=torch.float32, device=x.device)
    out = torch.empty((graph.n_rows, h), dtype=torch.float32, device=x.device)
    side.wait_stream(main)  # the buffers above may recycle blocks whose last use is still queued on the caller's stream
    for i, (c0, c1) in enumerate(slices):
        kernels.gemm(x, w[:, c0:c1], out=support[:, c0:c1], tile=PIPELINE_TILE)
        ready = torch.cuda.Event()
        ready.record(main)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            kernels.spmm_csr_relu(graph.rowptr, graph.col, graph.val, support, n_cols=graph.n_cols, act=kernels.ACT_RELU, out_mask=mask,
                                  out=out, slices=(c0 // 128, c1 // 128), resident=PIPELINE_RESIDENT if i + 1 < len(slices) else None,
                                  tag="spmm_csr_f32[fwd]")
    main.wait_stream(side)
    return out


def _pipelined_backward(x, dy, gt, mask, slices):
    """dS = A^T (dy * [Y > 0]) and dW = X^T dS, the gather of slice c + 1 on the side stream next to the GEMM of slice c."""
    main, side = torch.cuda.current_stream(x.device), _side_stream(x.device)
    h = dy.shape[1]
    ds = torch.empty((gt.n_rows, h), dtype=torch.float32, device=x.device)
    dw = torch.empty((x.shape[1], h), dtype=torch.float32, device=x.device)
    side.wait_stream(main)  # dy (and the recycled blocks of ds / dw) are ready on the caller's stream
    for i, (c0, c1) in enumerate(slices):
        with torch.cuda.stream(side):
            kernels.spmm_csr_relu(gt.rowptr, gt.col, gt.val, dy, n_cols=gt.n_cols, in_mask=mask, out=ds, slices=(c0 // 128, c1 // 128),
                                  resident=PIPELINE_RESIDENT if i > 0 else None, tag="spmm_csr_f32[bwd]")
            ready = torch.cuda.Event()
            ready.record(side)
        main.wait_event(ready)
        kernels.gemm(x, ds[:, c0:c1], trans_a=True, out=dw[:, c0:c1], tile=PIPELINE_TILE)
    main.wait_stream(side)
    return ds, dw


class _GCNLayerFn(torch.autograd.Function):
    """out = act(rowscale * reduce_e(val_e * colscale[src] * (x W)[src]) + bias); scales / mean are optional."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], graph: CSRGraph,
                active: bool, rowscale: Optional[torch.Tensor], colscale: Optional[torch.Tensor], reduce: int):
        x = x.contiguous() if x.stride(-1) != 1 else x
        w = weight.contiguous()
        ctx.narrow = False
        if (NARROW_FUSED and rowscale is None and colscale is None and reduce == kernels.REDUCE_SUM and graph.n_rows >= NARROW_MIN_ROWS
                and kernels.gcn_narrow_supported(w.shape[0], w.shape[1]) and x.stride(0) % 2 == 0 and x.data_ptr() % 8 == 0):
            # narrow layer (in < 64, out <= 64): aggregate-first, one gather kernel forward, one streaming kernel for dW / db
            out, agg = kernels.gcn_narrow_forward(graph.rowptr, graph.col, graph.val, x, w, bias, kernels.ACT_RELU if active else kernels.ACT_NONE,
                                                  n_cols=graph.n_cols, want_agg=weight.requires_grad or (bias is not None and bias.requires_grad))
            ctx.narrow, ctx.slices = True, None
            ctx.graph, ctx.active, ctx.has_bias = graph, active, bias is not None
            ctx.rowscale, ctx.colscale, ctx.reduce = None, None, reduce
            ctx.save_for_backward(x, w, out if active else None, agg)
            return out
        # ReLU fused into both SpMMs (sign mask instead of G = dY*[Y>0] in HBM) for the plain wide-layer case
        mask = None
        fused = (active and bias is None and rowscale is None and colscale is None and reduce == kernels.REDUCE_SUM
                 and w.shape[1] % 4 == 0 and kernels.relu_mask_bytes(graph.n_rows, w.shape[1]) > 0)
        slices = pipeline_slices(graph.n_rows, w.shape[1]) if (fused and kernels.GEMM_MODE == "exact" and x.is_cuda) else None
        ctx.slices = slices
        if slices is not None:
            mask = torch.empty(kernels.relu_mask_bytes(graph.n_rows, w.shape[1]), dtype=torch.uint8, device=x.device)
            out = _pipelined_forward(x, w, graph, mask, slices)
            ctx.graph, ctx.active, ctx.has_bias = graph, active, False
            ctx.rowscale, ctx.colscale, ctx.reduce = None, None, reduce
            ctx.save_for_backward(x, w, None, mask)
            return out
        support = kernels.gemm(x, w)
        if fused:
            mask = torch.empty(kernels.relu_mask_bytes(graph.n_rows, support.shape[1]), dtype=torch.uint8, device=x.device)
            out = kernels.spmm_csr_relu(graph.rowptr, graph.col, graph.val, support, n_cols=graph.n_cols,
                                        act=kernels.ACT_RELU, out_mask=mask, tag="spmm_csr_f32[fwd]")
        else:
            out = kernels.spmm_csr(graph.rowptr, graph.col, graph.val, support, n_cols=graph.n_cols, bias=bias,
                                   rowscale=rowscale, colscale=colscale, reduce=reduce,
                                   act=kernels.ACT_RELU if active else kernels.ACT_NONE, tag="spmm_csr_f32[fwd]")
        ctx.graph, ctx.active, ctx.has_bias = graph, active, bias is not None
        ctx.rowscale, ctx.colscale, ctx.reduce = rowscale, colscale, reduce
        ctx.save_for_backward(x, w, out if (active and not fused) else None, mask)
        return out

    @staticmethod
    def backward(ctx, dy: torch.Tensor):
        x, w, out, mask = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dy = dy.contiguous() if dy.stride(-1) != 1 else dy
        dx = dw = db = None
        if ctx.narrow:
            agg = mask  # fourth saved slot: the aggregated rows [n, 64]
            if need_w or need_b:
                dw, db = kernels.gcn_narrow_backward(agg, dy, w.shape[0], y_act=out if ctx.active else None, want_bias=need_b)
                dw = dw if need_w else None
            if need_x:  # d x = A^T ((dy * [y > 0]) W^T): the generic kernels at width `in`
                g = kernels.relu_backward(out, dy) if ctx.active else dy
                gt = ctx.graph.transpose()
                dx = kernels.spmm_csr(gt.rowptr, gt.col, gt.val, kernels.gemm(g, w, trans_b=True), n_cols=gt.n_cols, tag="spmm_csr_f32[bwd]")
            return dx, dw, db, None, None, None, None, None
        if mask is not None:  # fused ReLU backward: dS = A^T (dy * [out > 0]) in one gather
            if need_x or need_w:
                gt = ctx.graph.transpose()
                if dy.stride(0) % 4 != 0 or dy.data_ptr() % 16 != 0:
                    dy = dy.clone(memory_format=torch.contiguous_format)  # fresh allocation: 16-byte aligned rows
                if ctx.slices is not None and need_w and not need_x:
                    _, dw = _pipelined_backward(x, dy, gt, mask, ctx.slices)
                    return None, dw, None, None, None, None, None, None
                if BWD_MASK_MODE == "premask":
                    g = kernels.relu_mask_apply(dy, mask)
                    ds = kernels.spmm_csr(gt.rowptr, gt.col, gt.val, g, n_cols=gt.n_cols, tag="spmm_csr_f32[bwd]")
                    del g
                else:
                    ds = kernels.spmm_csr_relu(gt.rowptr, gt.col, gt.val, dy, n_cols=gt.n_cols, in_mask=mask, tag="spmm_csr_f32[bwd]")
                if need_w:
                    dw = kernels.gemm(x, ds, trans_a=True)
                if need_x:
                    dx = kernels.gemm(ds, w, trans_b=True)
            return dx, dw, db, None, None, None, None, None
        g = kernels.relu_backward(out, dy) if ctx.active else dy
        if need_b:
            db = kernels.colsum(g)
        if need_x or need_w:
            gt = ctx.graph.transpose()
            # d(x W)[u] = colscale[u] * sum_{e: u -> v} val_e * m[v] * g[v],  m = rowscale (/ in-degree for mean)
            m = ctx.rowscale
            if ctx.reduce == kernels.REDUCE_MEAN:
                deg = (ctx.graph.rowptr[1:] - ctx.graph.rowptr[:-1]).to(torch.float32).clamp(min=1)
                m = (1.0 / deg) if m is None else m / deg
            ds = kernels.spmm_csr(gt.rowptr, gt.col, gt.val, g, n_cols=gt.n_cols, rowscale=ctx.colscale, colscale=m,
                                  tag="spmm_csr_f32[bwd]")
            if need_w:
                dw = kernels.gemm(x, ds, trans_a=True)
            if need_x:
                dx = kernels.gemm(ds, w, trans_b=True)
        return dx, dw, db, None, None, None, None, None


def gcn_layer(x: torch.Tensor, weight: torch.Tensor, graph: CSRGraph, bias: Optional[torch.Tensor] = None,
              active: bool = False, *, rowscale: Optional[torch.Tensor] = None, colscale: Optional[torch.Tensor] = None,
              reduce: int = kernels.REDUCE_SUM) -> torch.Tensor:
    """act(rowscale * reduce(A diag(colscale) (x @ weight)) + bias) with a hand-written HIP forward and backward."""
    return _GCNLayerFn.apply(x, weight, bias, graph, active, rowscale, colscale, reduce)


class _SpMMFn(torch.autograd.Function):
    """y = rowscale * reduce_e(val_e * colscale[src] * x[src]) — the aggregation alone (no dense transform), e.g. HetConv of
    scHeteroNet (scheteronet.py:383-386) and the energy propagation (:611-640).  Backward = the same kernel on the cached
    CSR of A^T with the two scale vectors swapped."""

    @staticmethod
    def forward(ctx, x, graph: CSRGraph, rowscale, colscale, reduce: int):
        x = x.contiguous()
        ctx.graph, ctx.rowscale, ctx.colscale, ctx.reduce = graph, rowscale, colscale, reduce
        return kernels.spmm_csr(graph.rowptr, graph.col, graph.val, x, n_cols=graph.n_cols, rowscale=rowscale, colscale=colscale, reduce=reduce)

    @staticmethod
    def backward(ctx, dy):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        gt = ctx.graph.transpose()
        m = ctx.rowscale
        if ctx.reduce == kernels.REDUCE_MEAN:
            deg = (ctx.graph.rowptr[1:] - ctx.graph.rowptr[:-1]).to(torch.float32).clamp(min=1)
            m = (1.0 / deg) if m is None else m / deg
        dx = kernels.spmm_csr(gt.rowptr, gt.col, gt.val, dy.contiguous(), n_cols=gt.n_cols, rowscale=ctx.colscale, colscale=m)
        return dx, None, None, None, None


def spmm(x: torch.Tensor, graph: CSRGraph, *, rowscale: Optional[torch.Tensor] = None, colscale: Optional[torch.Tensor] = None,
         reduce: int = kernels.REDUCE_SUM) -> torch.Tensor:
    """rowscale * reduce(A diag(colscale) x) on the CSR SpMM kernel, differentiable in x."""
    return _SpMMFn.apply(x, graph, rowscale, colscale, reduce)


class _GATAggregateFn(torch.autograd.Function):
    """out[i] = sum_{e=(j->i)} att[e] x[j],  att = edge_softmax(act(a_src[j] + a_dst[i]))  — GATConv.propagate of
    stagate.py:104-124 without the [E, C] messages: dh_edge_softmax_f32 + dh_spmm_csr_f32 forward; backward from
    dh_sddmm_csr_f32 (d att), dh_edge_softmax_backward_f32 (softmax / activation Jacobians, d a_dst), a scatter-add over the
    source ids (d a_src) and the SpMM on the transposed CSR with the attention values carried along (d x)."""

    @staticmethod
    def forward(ctx, x, a_src, a_dst, graph: CSRGraph, act: int, slope: float, shift=None, edge_scale=None):
        x, a_src, a_dst = x.contiguous(), a_src.contiguous(), a_dst.contiguous()
        att = kernels.edge_softmax(graph.rowptr, graph.col, a_src, a_dst, act=act, negative_slope=slope, shift=shift)
        # edge_scale [E] (CSR order): dropout on the attention coefficients (0 or 1 / (1 - p)), applied between softmax and sum
        out = kernels.spmm_csr(graph.rowptr, graph.col, att if edge_scale is None else att * edge_scale, x, n_cols=graph.n_cols)
        ctx.graph, ctx.act, ctx.slope = graph, act, slope
        ctx.save_for_backward(x, a_src, a_dst, att, edge_scale)
        ctx.mark_non_differentiable(att)
        return out, att

    @staticmethod
    def backward(ctx, dout, _datt_unused):
        x, a_src, a_dst, att, edge_scale = ctx.saved_tensors
        g = ctx.graph
        dout = dout.contiguous()
        dx = da_src = da_dst = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            u, v = dout, x
            if x.shape[1] % 4:  # the SDDMM moves 16 bytes per lane: zero-pad the inner dimension (the products are unchanged)
                pad = 4 - x.shape[1] % 4
                u, v = torch.nn.functional.pad(dout, (0, pad)), torch.nn.functional.pad(x, (0, pad))
            datt = kernels.sddmm_csr(g.rowptr, g.col, u, v, scale=edge_scale)
            dt, da_dst = kernels.edge_softmax_backward(g.rowptr, g.col, a_src, a_dst, att, datt, act=ctx.act, negative_slope=ctx.slope)
            da_src = torch.zeros_like(a_src).index_add_(0, g.col.long(), dt)
        if ctx.needs_input_grad[0]:
            if getattr(g, "_t_perm", None) is None:  # structure of A^T + where each of its entries sits in A: once per graph
                g._t_struct = kernels.csr_transpose(g.rowptr, g.col, None, g.n_rows, g.n_cols)
                g._t_perm = g._t_struct[3].long()
            rp_t, col_t = g._t_struct[0], g._t_struct[1]
            val = att if edge_scale is None else att * edge_scale
            dx = kernels.spmm_csr(rp_t, col_t, val[g._t_perm].contiguous(), dout, n_cols=g.n_rows)
        return dx, da_src, da_dst, None, None, None, None, None


def gat_aggregate(x, a_src, a_dst, graph: CSRGraph, *, act: int = kernels.ATT_SIGMOID, negative_slope: float = 0.2, shift=None,
                  edge_scale=None):
    """(out, att): attention-weighted aggregation over the in-edges of every row and the per-edge coefficients (CSR order).
    ``shift``: one-element tensor subtracted from the scores before exp instead of each row's maximum (treated as a constant:
    its gradient is eps / (row sum + eps) of an attention, i.e. nothing)."""
    return _GATAggregateFn.apply(x, a_src, a_dst, graph, act, negative_slope, shift, edge_scale)


class _EdgeWeightedSumFn(torch.autograd.Function):
    """out[i] = sum_{e=(j->i)} w[e] x[j] with gradients for x AND for the per-edge weights w (CSR order): dh_spmm_csr_f32 forward, the
    SpMM over the transposed structure with the weights carried along for dx, dh_sddmm_csr_f32 for dw[e] = <dout[i], x[j]>."""

    @staticmethod
    def forward(ctx, x, w, graph: CSRGraph):
        x, w = x.contiguous(), w.contiguous()
        ctx.graph = graph
        ctx.save_for_backward(x, w)
        return kernels.spmm_csr(graph.rowptr, graph.col, w, x, n_cols=graph.n_cols)

    @staticmethod
    def backward(ctx, dout):
        x, w = ctx.saved_tensors
        g = ctx.graph
        dout = dout.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[1]:
            u, v = dout, x
            if x.shape[1] % 4:  # the SDDMM moves 16 bytes per lane: zero-pad the inner dimension (the products are unchanged)
                pad = 4 - x.shape[1] % 4
                u, v = torch.nn.functional.pad(dout, (0, pad)), torch.nn.functional.pad(x, (0, pad))
            dw = kernels.sddmm_csr(g.rowptr, g.col, u, v)
        if ctx.needs_input_grad[0]:
            if getattr(g, "_t_perm", None) is None:  # structure of A^T + where each of its entries sits in A: once per graph
                g._t_struct = kernels.csr_transpose(g.rowptr, g.col, None, g.n_rows, g.n_cols)
                g._t_perm = g._t_struct[3].long()
            dx = kernels.spmm_csr(g._t_struct[0], g._t_struct[1], w[g._t_perm].contiguous(), dout, n_cols=g.n_rows)
        return dx, dw, None


def edge_weighted_sum(x, w, graph: CSRGraph):
    """Aggregation with learnable per-edge weights (``w`` in CSR order of ``graph``): see _EdgeWeightedSumFn."""
    return _EdgeWeightedSumFn.apply(x, w, graph)


class _ZINBNLL(torch.autograd.Function):
    """``ZINBLoss.forward`` (dance/utils/loss.py:780-829) as two fused kernels instead of ~25 elementwise passes over the N x G
    matrices: dh_zinb_nll_forward_f32 (float64 row sums of the element loss) and dh_zinb_nll_backward_f32 (recomputes the element
    terms, writes d mean / d disp / d pi).  The result is a float64 scalar, as the reference's (its size factors are float64 and
    promote the expression)."""

    @staticmethod
    def forward(ctx, x, mean, disp, pi, scale_factor, ridge_lambda: float, logits: bool = False):
        x, mean, disp, pi = (t.contiguous() if t.stride(-1) != 1 else t for t in (x.float(), mean, disp, pi))
        sf = None if scale_factor is None else scale_factor.detach().to(torch.float64).contiguous()
        rowloss = kernels.zinb_nll_forward(x, mean, disp, pi, sf, ridge_lambda, logits=logits)
        ctx.ridge, ctx.logits = float(ridge_lambda), bool(logits)
        ctx.save_for_backward(x, mean, disp, pi, sf)
        return rowloss.sum() / float(x.shape[0] * x.shape[1])

    @staticmethod
    def backward(ctx, g):
        x, mean, disp, pi, sf = ctx.saved_tensors
        up = (g.to(torch.float64) / float(x.shape[0] * x.shape[1])).reshape(1).contiguous()
        dm, dd, dp = kernels.zinb_nll_backward(x, mean, disp, pi, sf, ctx.ridge, up, logits=ctx.logits)
        return None, dm, dd, dp, None, None, None


def zinb_nll(x, mean, disp, pi, scale_factor=None, ridge_lambda: float = 0.0) -> torch.Tensor:
    """mean over all (cell, gene) of the ZINB negative log-likelihood; float64 scalar (see _ZINBNLL)."""
    return _ZINBNLL.apply(x, mean, disp, pi, scale_factor, ridge_lambda, False)


def zinb_nll_from_logits(x, mean_raw, disp_raw, pi_raw, scale_factor=None, ridge_lambda: float = 0.0) -> torch.Tensor:
    """``zinb_nll(x, MeanAct(mean_raw), DispAct(disp_raw), sigmoid(pi_raw), ...)`` — the loss on the decoder heads of scdsc.py:409-411 /
    sctag.py — with the three activations and their backward evaluated inside the two loss kernels instead of as ~17 elementwise
    torch passes over the cells x genes matrices (dh_zinb_nll_logits_*)."""
    return _ZINBNLL.apply(x, mean_raw, disp_raw, pi_raw, scale_factor, ridge_lambda, True)


class _MixFn(torch.autograd.Function):
    """``a * x + b * y`` as one kernel (forward) and one per operand that needs a gradient (backward): see ``mix``."""

    @staticmethod
    def forward(ctx, x, y, a: float, b: float):
        ctx.a, ctx.b = float(a), float(b)
        return kernels.axpby(a, x if x.stride(-1) == 1 else x.contiguous(), b, y if y.stride(-1) == 1 else y.contiguous())

    @staticmethod
    def backward(ctx, g):
        g = g if g.stride(-1) == 1 else g.contiguous()
        return (kernels.axpby(ctx.a, g) if ctx.needs_input_grad[0] else None, kernels.axpby(ctx.b, g) if ctx.needs_input_grad[1] else None,
                None, None)


def mix(x, y, a: float, b: float):
    """``a * x + b * y`` for two fp32 matrices — scDSC's ``(1 - sigma) * h + sigma * tra`` (scdsc.py:454-459) — with the reference
    expression's roundings (each product, then the sum) in one pass instead of three kernels and two temporaries."""
    if x.dim() != 2 or x.shape != y.shape or x.dtype != torch.float32 or y.dtype != torch.float32:
        return a * x + b * y
    return _MixFn.apply(x, y, a, b)


class _ZINBHeadsFn(torch.autograd.Function):
    """The three ZINB heads of scdsc.py:409-411 (``Linear`` + MeanAct / DispAct / Sigmoid on one hidden matrix) AND the loss on them
    (dance/utils/loss.py:780-829) as one function of (h, the heads' weights and biases): three exact-fp32 products, then ONE pass
    (dh_zinb_heads_fused_f32) that evaluates every element once — loss, the gradients w.r.t. the raw outputs written over them, and
    their column sums (the bias gradients).  ``zinb_nll_from_logits`` behind three ``HipLinear`` reads the N x G operands three times
    (loss, gradient, bias column sums: 30 ms of a 201 ms epoch at 1M x 2000) for the same numbers.  The element gradients are
    computed for the mean's constant 1 / (N G); the run-time upstream scalar multiplies the small results (dW [G, H], db [G], and dh)
    in backward, so nothing N x G is touched again."""

    @staticmethod
    def forward(ctx, h, wm, bm, wd, bd, wp, bp, x, scale_factor, ridge_lambda: float):
        h = h.contiguous()
        x = x.float()
        x = x.contiguous() if x.stride(-1) != 1 else x
        n, g = x.shape
        raws = [kernels.gemm(h, w.contiguous(), trans_b=True, bias=None if b is None else b.detach()) for w, b in ((wm, bm), (wd, bd), (wp, bp))]
        sf = None if scale_factor is None else scale_factor.detach().to(torch.float64).contiguous()
        total, db = kernels.zinb_heads_fused_(x, raws[0], raws[1], raws[2], sf, ridge_lambda, 1.0 / float(n * g))
        ctx.save_for_backward(h, wm, wd, wp, db, *raws)
        ctx.has_bias = (bm is not None, bd is not None, bp is not None)
        return total / float(n * g)

    @staticmethod
    def backward(ctx, gout):
        h, wm, wd, wp, db, dm, dd, dp = ctx.saved_tensors
        gf = gout.to(torch.float32)
        need = ctx.needs_input_grad
        dh = None
        if need[0]:
            dh = kernels.gemm(dm, wm.contiguous())
            kernels.gemm(dd, wd.contiguous(), out=dh, accumulate=True)
            kernels.gemm(dp, wp.contiguous(), out=dh, accumulate=True)
            dh = dh * gf
        grads = [dh]
        for i, d in enumerate((dm, dd, dp)):
            grads.append(kernels.gemm(d, h, trans_a=True) * gf if need[1 + 2 * i] else None)
            grads.append(db[i] * gf if ctx.has_bias[i] and need[2 + 2 * i] else None)
        return (*grads, None, None, None)


def zinb_heads_loss(h, heads, x, scale_factor=None, ridge_lambda: float = 0.0) -> torch.Tensor:
    """``zinb_nll_from_logits(x, L_mean(h), L_disp(h), L_pi(h), ...)`` for three ``nn.Linear``-like ``heads`` (weight [G, H], bias [G] or
    None), evaluated by _ZINBHeadsFn: float64 scalar, gradients for ``h`` and the heads' parameters."""
    (lm, ld_, lp) = heads
    return _ZINBHeadsFn.apply(h, lm.weight, lm.bias, ld_.weight, ld_.bias, lp.weight, lp.bias, x, scale_factor, ridge_lambda)


class _SoftmaxXentSum(torch.autograd.Function):
    """``F.cross_entropy(logits, labels, reduction="sum")`` as one kernel + a scalar multiply in backward (dh_softmax_xent_sum_f32)."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index: int):
        x = logits if logits.dtype == torch.float32 and logits.stride(-1) == 1 else logits.float().contiguous()
        need = logits.requires_grad
        loss, d = kernels.softmax_xent_sum(x, labels, ignore_index, want_grad=need)
        ctx.in_dtype = logits.dtype
        if need:
            ctx.save_for_backward(d)
        return loss

    @staticmethod
    def backward(ctx, g):
        d, = ctx.saved_tensors
        return (d * g).to(ctx.in_dtype), None, None


class CrossEntropySum(torch.nn.Module):
    """``nn.CrossEntropyLoss(reduction="sum")`` (scdeepsort.py:185) on dh_softmax_xent_sum_f32 for 2-D logits with int64 class labels on
    the device; class probabilities as targets or more dimensions are torch's."""

    def __init__(self, ignore_index: int = -100):
        super().__init__()
        self.ignore_index = int(ignore_index)

    def forward(self, logits, labels):
        if logits.dim() == 2 and labels.dim() == 1 and labels.dtype == torch.int64:
            return _SoftmaxXentSum.apply(logits, labels, self.ignore_index)
        return torch.nn.functional.cross_entropy(logits, labels, reduction="sum", ignore_index=self.ignore_index)


class _AdjReconstructionMSE(torch.autograd.Function):
    """mean_ij (sigmoid(<z_i, z_j>) - a_ij)^2 over ALL n^2 pairs for a sparse target a (CSR, stored entries only), as a function of z
    and without any n x n matrix — scTAG's adjacency-decoder loss ``F.mse_loss(sigmoid(z0 z0^T), adj)`` (sctag.py:470-471, :254):

        sum_ij (s_ij - a_ij)^2 = sum_ij s_ij^2  -  2 sum_{(i,j) in A} a_ij s_ij  +  sum_A a_ij^2,      s = sigmoid(z z^T)

    The first sum and its gradient 2 sum_j 2 s^2 (1 - s) z_j are one pass on the fp32 matrix cores that keeps every logit tile in
    registers (dh_gram_pairwise_f32, DH_GRAM_SIGMOID_SQ); the second needs the logits of the stored entries only (dh_sddmm_csr_f32)
    and its gradient is an SpMM over A and one over A^T with the per-edge coefficients as values."""

    @staticmethod
    def forward(ctx, z, graph: CSRGraph):
        z = z.contiguous()
        n = z.shape[0]
        rowloss, o = kernels.gram_pairwise(z, kernels.GRAM_SIGMOID_SQ)
        u, v = z, z
        if z.shape[1] % 4:  # the SDDMM moves 16 bytes per lane
            u = v = torch.nn.functional.pad(z, (0, 4 - z.shape[1] % 4))
        xe = kernels.sddmm_csr(graph.rowptr, graph.col, u, v)
        a = graph.val if graph.val is not None else torch.ones_like(xe)
        se = torch.sigmoid(xe)
        total = rowloss.sum(dtype=torch.float64) + (a * (a - 2 * se)).sum(dtype=torch.float64)
        ctx.graph = graph
        ctx.save_for_backward(z, o, se, a)
        return (total / float(n) / float(n)).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        z, o, se, a = ctx.saved_tensors
        graph = ctx.graph
        n = z.shape[0]
        scale = (g / float(n) / float(n)).to(torch.float32)
        ce = (-2.0 * a * se * (1 - se)).contiguous()  # d/dx of -2 a sigmoid(x); entry (i, j) contributes ce z_j to row i and ce z_i to row j
        dz = 2.0 * o + kernels.spmm_csr(graph.rowptr, graph.col, ce, z, n_cols=graph.n_cols)
        if graph.symmetric:  # same pattern and, for a symmetric target, the same coefficients: A^T's pass is A's
            dz = dz + kernels.spmm_csr(graph.rowptr, graph.col, ce, z, n_cols=graph.n_cols)
        else:
            if getattr(graph, "_t_perm", None) is None:
                graph._t_struct = kernels.csr_transpose(graph.rowptr, graph.col, None, graph.n_rows, graph.n_cols)
                graph._t_perm = graph._t_struct[3].long()
            dz = dz + kernels.spmm_csr(graph._t_struct[0], graph._t_struct[1], ce[graph._t_perm].contiguous(), z, n_cols=graph.n_rows)
        return dz * scale, None


def adj_reconstruction_mse(z: torch.Tensor, graph: CSRGraph) -> torch.Tensor:
    """mean((sigmoid(z z^T) - A)^2) over all n^2 entries for a sparse A, in O(n d) memory (see _AdjReconstructionMSE)."""
    return _AdjReconstructionMSE.apply(z, graph)


class _DenseAdjLayerFn(torch.autograd.Function):
    """Same op for a DENSE adjacency (SpaGCN passes a dense N x N FloatTensor, spagcn.py:497,359)."""

    @staticmethod
    def forward(ctx, x, weight, bias, adj):
        x = x.contiguous() if x.stride(-1) != 1 else x
        w = weight.contiguous()
        adj = adj.contiguous() if adj.stride(-1) != 1 else adj
        support = kernels.gemm(x, w)
        out = kernels.gemm(adj, support)
        if bias is not None:
            kernels.bias_act_(out, bias)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w, adj)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, w, adj = ctx.saved_tensors
        dy = dy.contiguous() if dy.stride(-1) != 1 else dy
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = kernels.colsum(dy)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            ds = kernels.gemm(adj, dy, trans_a=True)
            if ctx.needs_input_grad[1]:
                dw = kernels.gemm(x, ds, trans_a=True)
            if ctx.needs_input_grad[0]:
                dx = kernels.gemm(ds, w, trans_b=True)
        return dx, dw, db, None


def dense_adj_layer(x, weight, adj, bias=None):
    return _DenseAdjLayerFn.apply(x, weight, bias, adj)


class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) (torch.nn.Linear semantics, optional fused ReLU) on the matrix cores.

    fp32 input: dh_gemm_f32_bias_act (exact fp32 MFMA, bias / ReLU in the epilogue).  bf16 input (config C3): dh_gemm_bf16 with the bias /
    ReLU epilogue fused; W (an fp32 master parameter) is rounded to bf16 for the products, dW / db come back in fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, out_dtype):
        x = x.contiguous()
        act = kernels.ACT_RELU if relu else kernels.ACT_NONE
        if x.dtype == torch.bfloat16:
            w = weight.detach().to(torch.bfloat16).contiguous()
            y = kernels.gemm_bf16(x, w, trans_b=True, bias=None if bias is None else bias.detach().float(), act=act,
                                  out_dtype=out_dtype or torch.bfloat16)
        else:
            w = weight.contiguous()
            y = kernels.gemm(x, w, trans_b=True, bias=None if bias is None else bias.detach(), act=act)  # bias / ReLU in the tile's store
        ctx.has_bias, ctx.relu = bias is not None, relu
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if x.dtype == torch.bfloat16:
            g = dy.to(torch.bfloat16)
            if ctx.relu:
                g = kernels.relu_backward_bf16(y if y.dtype == torch.bfloat16 else y.to(torch.bfloat16), g)
            if ctx.needs_input_grad[0]:
                dx = kernels.gemm_bf16(g, w)
            if ctx.needs_input_grad[1]:
                dw = kernels.gemm_bf16(g, x, trans_a=True, out_dtype=torch.float32)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = kernels.colsum_bf16(g)
        else:
            g = kernels.relu_backward(y, dy) if ctx.relu else dy
            if ctx.needs_input_grad[0]:
                dx = kernels.gemm(g, w)
            if ctx.needs_input_grad[1]:
                dw = kernels.gemm(g, x, trans_a=True)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = kernels.colsum(g)
        return dx, dw, db, None, None


def linear(x, weight, bias=None, *, relu: bool = False, out_dtype=None):
    return _LinearFn.apply(x, weight, bias, relu, out_dtype)


class HipLinear(torch.nn.Linear):
    """``nn.Linear`` (same parameters / state_dict) whose forward and backward run on dh_gemm_f32, or on
    dh_gemm_bf16 when the input is a bf16 tensor.  ``fuse_relu`` folds a following ``nn.ReLU`` into the epilogue;
    ``out_dtype`` (bf16 inputs only) selects fp32 output, e.g. for the logits of the last layer."""

    out_dtype = None

    def forward(self, input, fuse_relu: bool = False):
        return linear(input, self.weight, self.bias, relu=fuse_relu, out_dtype=self.out_dtype)


class _StudentTFn(torch.autograd.Function):
    """q = normalise_j((1 / ((1 + ||z_i - mu_j||^2 / a) + eps))^pw * scale): the DEC heads' soft assignment on the fused kernel pair
    dh_student_t_forward_f32 / dh_student_t_backward_f32 (csrc/student_t.hip) — no [N, C, d] broadcast tensor, forward or backward."""

    @staticmethod
    def forward(ctx, z, mu, a, eps, pw, scale):
        z = z.contiguous() if z.stride(-1) != 1 else z
        mu = mu.contiguous()
        ctx.consts = (float(a), float(eps), float(pw), float(scale))
        ctx.save_for_backward(z, mu)
        return kernels.student_t_forward(z, mu, *ctx.consts)

    @staticmethod
    def backward(ctx, dq):
        z, mu = ctx.saved_tensors
        if not (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            return None, None, None, None, None, None
        dq = dq.contiguous() if dq.stride(-1) != 1 else dq
        dz, dmu = kernels.student_t_backward(z, mu, *ctx.consts, dq, want_dz=ctx.needs_input_grad[0])
        return dz, (dmu if ctx.needs_input_grad[1] else None), None, None, None, None


class _DecKLFn(torch.autograd.Function):
    """scale * sum_ij p log(p / (q + eps)), differentiable in q (p is the fixed target): spagcn.py:398-407 on dh_dec_kl_forward / _backward_f32."""

    @staticmethod
    def forward(ctx, p, q, eps, scale):
        p = p if p.stride(-1) == 1 else p.contiguous()
        q = q if q.stride(-1) == 1 else q.contiguous()
        ctx.save_for_backward(p, q)
        ctx.consts = (float(eps), float(scale))
        return kernels.dec_kl_forward(p, q, eps, scale)

    @staticmethod
    def backward(ctx, g):
        p, q = ctx.saved_tensors
        return None, kernels.dec_kl_backward(p, q, *ctx.consts, g.contiguous().to(torch.float32)), None, None


def dec_kl_loss(p: torch.Tensor, q: torch.Tensor, *, eps: float = 1e-6, scale: float) -> torch.Tensor:
    """``scale * sum(p * log(p / (q + eps)))`` — the DEC heads' KL loss (``scale`` = 1 / n_spots gives the reference's mean over spots of the
    row sums).  Fused kernels for fp32 device tensors with at most 4096 clusters; the reference's torch expression otherwise."""
    if p.is_cuda and q.is_cuda and p.dtype == torch.float32 and q.dtype == torch.float32 and p.dim() == 2 and p.shape == q.shape and q.shape[1] <= 4096:
        return _DecKLFn.apply(p.detach(), q, eps, scale)
    return torch.sum(p * torch.log(p / (q + eps))) * scale


def dec_target_distribution(q: torch.Tensor, colsum_q=None) -> torch.Tensor:
    """``p = q**2 / q.sum(0); p / p.sum(1, keepdim=True)`` (spagcn.py:421-425) — one kernel after the column sums for fp32 device tensors."""
    if q.is_cuda and q.dtype == torch.float32 and q.dim() == 2 and q.shape[1] <= 4096 and not q.requires_grad:
        return kernels.dec_target(q, colsum_q)
    f = torch.sum(q, dim=0) if colsum_q is None else colsum_q
    p = q**2 / f
    return p / torch.sum(p, dim=1, keepdim=True)


def student_t_assign(z: torch.Tensor, mu: torch.Tensor, *, a: float, eps: float, pw: float, scale: float = 1.0) -> torch.Tensor:
    """Soft cluster assignment of the DEC heads (SimpleGCDEC / GC_DEC: spagcn.py:394-396,605-607; ScDSCModel: scdsc.py:466-468), rows
    normalised, differentiable in ``z`` and ``mu``.  Shapes the fused kernels do not cover (kernels.student_t_supported: more than 64 clusters, C d > 4096, wide embeddings; non-fp32)
    take the reference's broadcast formulation in torch — same arithmetic, three tensors of N x C x d."""
    if z.dtype == torch.float32 and mu.dtype == torch.float32 and kernels.student_t_supported(mu.shape[0], z.shape[1]):
        return _StudentTFn.apply(z, mu, a, eps, pw, scale)
    q = 1.0 / ((1.0 + torch.sum((z.unsqueeze(1) - mu)**2, dim=2) / a) + eps)
    q = q**pw * scale
    return q / torch.sum(q, dim=1, keepdim=True)
