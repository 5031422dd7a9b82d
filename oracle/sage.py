"""Oracle for the DGL message-passing layers (test infrastructure; see oracle/__init__.py).

PARITY UNPINNED by reference output (dgl 1.1.3 is not installable here).  Restates
  * AdaptiveSAGE.message_func / forward      dance/models/nn/gnn.py:62-96
  * WeightedGraphConv.forward               dance/modules/single_modality/clustering/graphsc.py:428-484
on explicit edge lists (src, dst, weight) in numpy; DGL semantics used: fn.mean = sum over in-edges / in-degree
with 0 for isolated nodes, fn.sum, block dst nodes are the first num_dst src nodes [3P-memory].
"""
import numpy as np


def sage_alpha_index(src_cell_id, dst_cell_id, n_genes):
    """gnn.py:72-76 — which alpha entry scales each edge."""
    src_id, dst_id = np.asarray(src_cell_id), np.asarray(dst_cell_id)
    idx = np.full(src_id.shape, n_genes + 1, dtype=np.int64)  # :72 default = cell self loop
    idx = np.where((src_id >= 0) & (dst_id < 0), src_id, idx)  # :73 gene -> cell
    idx = np.where((dst_id >= 0) & (src_id < 0), dst_id, idx)  # :74 cell -> gene
    idx = np.where((dst_id >= 0) & (src_id >= 0), n_genes, idx)  # :75 gene self loop
    return idx


def sage_neigh(src, dst, weight, cell_id_src, cell_id_dst, alpha, h_src, n_dst):
    """neigh = mean over in-edges of (h_src * alpha[idx]) * w   (gnn.py:81-82,90), float32 per-edge products,
    float64 accumulation for a summation-order-independent check value."""
    alpha = np.asarray(alpha, dtype=np.float32).reshape(-1)
    idx = sage_alpha_index(cell_id_src[src], cell_id_dst[dst], alpha.shape[0] - 2)
    m = (h_src[src].astype(np.float32) * alpha[idx][:, None]).astype(np.float32)  # :81
    m = (m * np.asarray(weight, dtype=np.float32).reshape(-1, 1)).astype(np.float32)  # :82
    out = np.zeros((n_dst, h_src.shape[1]), dtype=np.float64)
    np.add.at(out, dst, m.astype(np.float64))
    deg = np.bincount(dst, minlength=n_dst)
    return (out / np.maximum(deg, 1)[:, None]).astype(np.float32)


def sage_alpha_grad(src, dst, weight, cell_id_src, cell_id_dst, n_genes, h_src, dneigh):
    """d loss / d alpha for loss = <neigh, dneigh> (the K7 gradient of the intended model)."""
    idx = sage_alpha_index(cell_id_src[src], cell_id_dst[dst], n_genes)
    deg = np.bincount(dst, minlength=dneigh.shape[0])
    dots = (h_src[src].astype(np.float64) * dneigh[dst].astype(np.float64)).sum(1)
    vals = np.asarray(weight, dtype=np.float64).reshape(-1) * dots / np.maximum(deg[dst], 1)
    return np.bincount(idx, weights=vals, minlength=n_genes + 2)


def weighted_graph_conv(src, dst, edge_weight, feat_src, n_dst, weight, bias=None, agg="sum", activation=None):
    """graphsc.py:428-484 with DGL GraphConv defaults norm='both', weight, bias [3P-memory]:
    feat_src * out_deg^-1/2 (:444-449, degrees clamped >= 1, computed on the block) -> @ W (:459-460) ->
    sum|mean of h_src * w_e (:462-465) -> * in_deg^-1/2 (:467-476) -> + bias -> activation."""
    n_src = feat_src.shape[0]
    out_deg = np.maximum(np.bincount(src, minlength=n_src), 1).astype(np.float32)
    h = feat_src.astype(np.float32) * (out_deg**-0.5)[:, None]
    h = (h.astype(np.float64) @ np.asarray(weight, dtype=np.float64))
    m = h[src] * np.asarray(edge_weight, dtype=np.float64).reshape(-1, 1)
    rst = np.zeros((n_dst, h.shape[1]), dtype=np.float64)
    np.add.at(rst, dst, m)
    in_deg_raw = np.bincount(dst, minlength=n_dst)
    if agg == "mean":
        rst = rst / np.maximum(in_deg_raw, 1)[:, None]
    in_deg = np.maximum(in_deg_raw, 1).astype(np.float64)
    rst = rst * (in_deg**-0.5)[:, None]
    if bias is not None:
        rst = rst + np.asarray(bias, dtype=np.float64)
    if activation == "relu":
        rst = np.maximum(rst, 0)
    return rst.astype(np.float32)
