"""GraphSCI's gene-graph network on MI355X — drop-in for ``GNNModel`` of
dance/modules/single_modality/imputation/graphsci.py:107-124: three ``dgl.nn.GraphConv`` layers (norm="both", tanh / relu) and the
two decoders over the gene-gene graph.  ``GraphConv`` here is DGL's layer (parameters ``weight`` [in, out] xavier-uniform,
``bias`` zeros; out = D_in^-1/2 A D_out^-1/2 X W + b with degrees clamped at 1) as ONE fused GCN layer op: the two degree
factors ride the SpMM as row / column scales (autograd.gcn_layer), the activation's ReLU in its epilogue.

``forward(g)``: ``g`` is a ``CSRGraph`` (rows = destination genes) carrying ``g.ndata["feat"]``, which is what the
FeatureFeatureGraph of the reference holds in its DGL graph.  The reference's quirk of computing ``z_adj_log_std`` with
``dec_mean`` (:121) is kept."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....autograd import gcn_layer
from ....graph import CSRGraph


class GraphConv(nn.Module):
    """``dgl.nn.GraphConv`` [3P-memory: dgl 1.1.3]: norm in {"none", "both", "right", "left"}, optional edge weights."""

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

    @staticmethod
    def _scales(graph: CSRGraph, norm: str):
        cache = graph.__dict__.setdefault("_graphconv_scales", {})
        if norm not in cache:
            in_deg = (graph.rowptr[1:] - graph.rowptr[:-1]).float().clamp(min=1)
            out_deg = torch.bincount(graph.col.long(), minlength=graph.n_cols).float().clamp(min=1)
            colscale = out_deg.pow(-0.5) if norm == "both" else (1.0 / out_deg if norm == "left" else None)
            rowscale = in_deg.pow(-0.5) if norm == "both" else (1.0 / in_deg if norm == "right" else None)
            cache[norm] = (colscale, rowscale)
        return cache[norm]

    def forward(self, graph: CSRGraph, feat, weight=None, edge_weight=None):
        if not self._allow_zero_in_degree and bool(((graph.rowptr[1:] - graph.rowptr[:-1]) == 0).any()):
            raise RuntimeError("There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                               "Adding self-loop on the input graph will resolve the issue.")
        if weight is not None and self.weight is not None:
            raise RuntimeError("External weight is provided while at the same time the module has defined its own weight parameter. "
                               "Please create the module with flag weight=False.")
        weight = self.weight if weight is None else weight
        colscale, rowscale = self._scales(graph, self._norm)
        if edge_weight is not None:
            g = CSRGraph(graph.rowptr, graph.col, edge_weight.reshape(-1).contiguous(), graph.n_rows, graph.n_cols)
        elif graph.val is None:
            g = graph
        else:  # DGL ignores stored edge data unless edge_weight is passed: the pattern-only view (and its transpose) is kept
            g = graph.__dict__.setdefault("_pattern", CSRGraph(graph.rowptr, graph.col, None, graph.n_rows, graph.n_cols, symmetric=False))
        relu = self._activation in (F.relu, torch.relu) or isinstance(self._activation, nn.ReLU)
        rst = gcn_layer(feat, weight, g, self.bias, relu, rowscale=rowscale, colscale=colscale)
        if self._activation is not None and not relu:
            rst = self._activation(rst)
        return rst


class GNNModel(nn.Module):

    def __init__(self, in_feats, out_feats, dropout=0., n_hidden1=256, n_hidden2=256):
        super().__init__()
        self.dp = nn.Dropout(dropout)
        self.conv1 = GraphConv(in_feats, n_hidden1, activation=nn.Tanh())
        self.conv2 = GraphConv(n_hidden1, n_hidden2, activation=nn.ReLU())
        self.dec_mean = GraphConv(n_hidden2, out_feats)
        self.dec_log_std = GraphConv(n_hidden2, out_feats)

    def forward(self, g):
        h = self.conv1(g, self.dp(g.ndata["feat"]))
        h = self.conv2(g, self.dp(h))
        z_adj_mean = self.dec_mean(g, self.dp(h))
        z_adj_log_std = self.dec_mean(g, self.dp(h))  # sic (graphsci.py:121)
        z_adj = torch.normal(z_adj_mean, torch.exp(z_adj_log_std))
        return z_adj, z_adj_log_std, z_adj_mean
