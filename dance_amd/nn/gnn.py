"""AdaptiveSAGE on MI355X — drop-in for dance/models/nn/gnn.py:8-96 (scDeepSort's weighted-mean SAGE layer).

``message_func`` + ``fn.mean`` (gnn.py:62-82,90) is one fused HIP kernel (dh_sage_aggregate_f32: the per-edge
alpha index is derived from the src/dst ``cell_id`` on the fly, no [E, D] message tensor is materialised).

PARITY NOTE.  The reference computes ``neigh`` and then ignores it: the layer output is
``norm(act(Linear(dropout(h_dst))))`` (gnn.py:92-96), so ``alpha`` receives no gradient.  We reproduce exactly that:
``neigh`` is still computed (it is where the reference spends its time) and kept in ``self.last_neigh`` for parity
checks; it does not enter the output unless ``use_neigh=True`` is set explicitly (the model the paper intends:
``z = h_dst + neigh``-style update is NOT silently substituted).
"""
import torch
import torch.nn as nn

from .. import kernels
from ..autograd import HipLinear


class _SageAggregateFn(torch.autograd.Function):
    """neigh = mean_e alpha[idx(e)] w_e h[src(e)] with gradients for h (transposed gather) and alpha (K7)."""

    @staticmethod
    def forward(ctx, h, alpha, block, differentiated=False):
        cid_src, cid_dst = block.srcdata["cell_id"], block.dstdata["cell_id"]
        win = getattr(block, "gene_window", None)
        # fp32 features whose aggregation enters the output (use_neigh=True) take the exact gather: the matrix-core kernel feeds
        # fp32 operands as bf16 hi + lo pairs (1e-5 relative per term), which is not the function the backward below differentiates
        exact = differentiated and h.dtype == torch.float32
        gr = getattr(block, "gene_rows", 0)
        if (gr and win is not None and kernels.SAGE_MODE == "mfma" and not exact and h.shape[1] % 4 == 0
                and kernels.sage_mfma_supported(win[1], h.shape[1], h.dtype)
                and kernels.sage_splitk_supported(gr, block.cell_window[1], h.shape[1], h.dtype, block.col.numel())):
            # every node is a destination (CellGeneGraph.all_rows_block): the gene rows take the cells as the K dimension of the
            # split-K matrix-core kernel, the cell rows the gene window — both without a dense adjacency, into one result
            hc, a32, rp = h.contiguous(), alpha.detach().float(), block.rowptr_dst
            neigh = torch.empty((block.number_of_dst_nodes(), h.shape[1]), dtype=h.dtype, device=h.device)
            kernels.sage_aggregate_splitk(rp[:gr + 1], block.col, block.val, cid_src, cid_dst[:gr], a32, hc, *block.cell_window, out=neigh[:gr])
            kernels.sage_aggregate_mfma(rp[gr:], block.col, block.val, cid_src, cid_dst[gr:], a32, hc, win[0], win[1], out=neigh[gr:])
        elif (not gr and win is not None and win[1] > 0 and kernels.SAGE_MODE == "mfma" and not exact and h.shape[1] % 4 == 0
                and kernels.sage_mfma_supported(win[1], h.shape[1], h.dtype)):
            # cell destinations whose gene rows are a known window of the sources: the matrix-core kernel (one launch)
            neigh = kernels.sage_aggregate_mfma(block.rowptr_dst, block.col, block.val, cid_src, cid_dst, alpha.detach().float(), h.contiguous(),
                                                win[0], win[1])
        else:
            agg = kernels.sage_aggregate_bf16 if h.dtype == torch.bfloat16 else kernels.sage_aggregate  # C3: bf16 storage
            neigh = agg(block.rowptr_dst, block.col, block.val, cid_src, cid_dst, alpha.detach().float(), h.contiguous())
        ctx.block = block
        ctx.save_for_backward(h, alpha)
        return neigh

    @staticmethod
    def backward(ctx, dneigh):
        h, alpha = ctx.saved_tensors
        blk = ctx.block
        if getattr(blk, "pad_row", False):
            raise NotImplementedError("the gradient of the aggregation (use_neigh=True) is not defined on a padded static block")
        cid_src, cid_dst = blk.srcdata["cell_id"], blk.dstdata["cell_id"]
        dneigh = dneigh.contiguous()
        n_genes = alpha.numel() - 2
        dalpha = dh = None
        bf16 = h.dtype == torch.bfloat16
        deg = (blk.rowptr[1:] - blk.rowptr[:-1]).to(torch.float32).clamp(min=1)
        rows = torch.repeat_interleave(torch.arange(blk.number_of_dst_nodes(), device=h.device),
                                       (blk.rowptr[1:] - blk.rowptr[:-1]).to(torch.int64), output_size=blk.col.numel())
        sid, did = cid_src[blk.col.to(torch.int64)], cid_dst[rows]
        idx = torch.full_like(sid, n_genes + 1, dtype=torch.int64)
        idx = torch.where((sid >= 0) & (did < 0), sid.to(torch.int64), idx)
        idx = torch.where((did >= 0) & (sid < 0), did.to(torch.int64), idx)
        idx = torch.where((did >= 0) & (sid >= 0), torch.full_like(idx, n_genes), idx)
        if ctx.needs_input_grad[1]:
            # dalpha[idx(e)] += w_e <H[src(e)], dneigh[dst(e)]> / deg(dst(e)): the per-edge dot products are an SDDMM
            # (dh_sddmm_csr_f32, fp32, fixed order); the (G + 2)-bin reduction runs in float64, where the order in which the
            # scatter-add visits the edges no longer reaches the fp32 result (dh_sage_alpha_grad_f32's fp32 atomics did)
            hf, df = h.float(), dneigh.float()
            if hf.shape[1] % 4:
                pad = 4 - hf.shape[1] % 4
                hf, df = torch.nn.functional.pad(hf, (0, pad)), torch.nn.functional.pad(df, (0, pad))
            c = kernels.sddmm_csr(blk.rowptr, blk.col, df.contiguous(), hf.contiguous(), scale=(blk.val / deg[rows]).contiguous())
            dalpha = torch.zeros(n_genes + 2, dtype=torch.float64, device=h.device).index_add_(0, idx, c.double()).float().reshape(alpha.shape)
        if ctx.needs_input_grad[0]:
            # dh[u] = sum_{e=(u->v)} alpha[idx(e)] w_e / deg(v) * dneigh[v]: gather over the transposed block
            ew = (alpha.reshape(-1)[idx] * blk.val / deg[rows]).contiguous()
            rp_t, col_t, val_t, _ = kernels.csr_transpose(blk.rowptr, blk.col, ew, blk.number_of_dst_nodes(),
                                                          blk.number_of_src_nodes())
            if bf16:
                dh = kernels.spmm_csr_bf16(rp_t, col_t, val_t, dneigh.to(torch.bfloat16), n_cols=blk.number_of_dst_nodes())
            else:
                dh = kernels.spmm_csr(rp_t, col_t, val_t, dneigh, n_cols=blk.number_of_dst_nodes())
        return dh, dalpha, None, None


class AdaptiveSAGE(nn.Module):

    def __init__(self, dim_in: int, dim_out: int, alpha: torch.Tensor, dropout_layer: nn.Module, act_layer: nn.Module,
                 norm_layer: nn.Module, *, use_neigh: bool = False, compute_neigh: bool = True):
        super().__init__()
        self.alpha = alpha
        self.gene_num = len(alpha) - 2
        self.use_neigh = use_neigh
        self.compute_neigh = compute_neigh
        self.last_neigh = None

        self.layers = nn.ModuleList()
        self.layers.append(dropout_layer)
        self.layers.append(HipLinear(dim_in, dim_out))
        nn.init.xavier_uniform_(self.layers[-1].weight, gain=nn.init.calculate_gain("relu"))
        self.layers.append(act_layer)
        self.layers.append(norm_layer)

    def aggregate(self, block, h):
        """``dstdata["neigh"]`` of the reference (gnn.py:90)."""
        return _SageAggregateFn.apply(h, self.alpha, block, torch.is_grad_enabled() and (h.requires_grad or self.alpha.requires_grad))

    def forward(self, block, h):
        off = getattr(block, "dst_offset", 0)  # 0 for sampled blocks (dst nodes lead the sources); G for the full-graph cell rows
        h_dst = h[off:off + block.number_of_dst_nodes()]
        if self.use_neigh:
            neigh = self.aggregate(block, h)
            self.last_neigh = neigh.detach()
            z = h_dst + neigh
        else:
            if self.compute_neigh:
                with torch.no_grad():  # computed and dropped, exactly like the reference (gnn.py:90-92)
                    self.last_neigh = self.aggregate(block, h.detach())
            z = h_dst
        dropout, lin, act, norm = self.layers
        z = dropout(z)
        if type(act) is nn.ReLU:  # the reference's configuration (scdeepsort.py:160): ReLU rides the GEMM epilogue
            z = lin(z, fuse_relu=True)
        else:
            z = act(lin(z))
        return norm(z)
