"""DSTG's graph convolution on MI355X — drop-in for ``GraphConvolution`` / ``GCN`` of
dance/modules/spatial/cell_type_deconvo/dstg.py:37-143 (parameters ``weight`` [in, out] glorot-uniform, optional ``bias``
U(+-1/sqrt(out)); ``forward(x, adj)`` = spmm(adj, x @ weight) (+ bias), x dense or sparse).  The layer is the fused GCN layer op
(dh_gemm_f32 + dh_spmm_csr_f32 with the bias in the SpMM epilogue, hand-written backward: autograd.gcn_layer)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from ....autograd import gcn_layer
from ....graph import CSRGraph, as_graph


class GraphConvolution(nn.Module):
    """Simple GCN layer, similar to https://arxiv.org/abs/1609.02907."""

    def __init__(self, in_features, out_features, support, bias=False):
        super().__init__()
        self.support = support
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        init_range = np.sqrt(6.0 / (self.in_features + self.out_features))  # glorot
        self.weight.data.uniform_(-init_range, init_range)
        stdv = 1. / np.sqrt(self.weight.size(1))
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, x, adj):
        if x.is_sparse:  # sparse node features (dstg.py:92-93): the feature GEMM wants them dense on the device
            x = x.to_dense()
        graph = adj if isinstance(adj, CSRGraph) else as_graph(adj, x.device)
        return gcn_layer(x, self.weight, graph, self.bias, False)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_features} -> {self.out_features})"


class GCN(nn.Module):
    """Dropout + GC + activation (dstg.py:108-143).  Note the reference passes ``bias`` in the ``support`` position of both
    layers (:114-115), so the layers never have a bias; kept."""

    def __init__(self, nfeat, nhid1, nout, bias=False, dropout=0., act=F.relu):
        super().__init__()
        self.gc1 = GraphConvolution(nfeat, nhid1, bias)
        self.gc2 = GraphConvolution(nhid1, nout, bias)
        self.dropout = dropout
        self.act = act

    def forward(self, x, adj):
        x = F.dropout(x, self.dropout, self.training) if not x.is_sparse else x
        x = self.gc1(x, adj)
        x = self.act(x)
        x = F.dropout(x, self.dropout, self.training)
        return self.gc2(x, adj)
