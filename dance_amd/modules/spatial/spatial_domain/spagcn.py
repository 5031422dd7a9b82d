"""SpaGCN's graph convolution on MI355X — drop-in for ``GraphConvolution`` of
dance/modules/spatial/spatial_domain/spagcn.py:337-366 (parameters ``weight``/``bias``, U(+-1/sqrt(out)) init,
``forward(input, adj)``).

The reference multiplies by a DENSE N x N adjacency (spagcn.py:497,359).  A dense ``adj`` tensor is honoured
with the MFMA GEMM (exact reference arithmetic, small N); a sparse tensor / ``CSRGraph`` (the kNN-truncated
Gaussian kernel used at scale) goes through the CSR SpMM.
"""
import numpy as np
import torch
from torch import nn

from ....autograd import dense_adj_layer, gcn_layer
from ....graph import CSRGraph, as_graph


class GraphConvolution(nn.Module):
    """Simple GCN layer, similar to https://arxiv.org/abs/1609.02907."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / np.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj):
        if isinstance(adj, torch.Tensor) and adj.layout == torch.strided:
            return dense_adj_layer(input, self.weight, adj, self.bias)
        graph = adj if isinstance(adj, CSRGraph) else as_graph(adj, input.device)
        return gcn_layer(input, self.weight, graph, self.bias, False)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_features} -> {self.out_features})"
