"""STdGCN's graph convolution on MI355X — drop-in for ``conGraphConvolutionlayer`` of
dance/modules/spatial/cell_type_deconvo/stdgcn.py:63-92 (``weight`` [in, out], ``bias`` [out], both U(+-1/sqrt(out));
``forward(input, adj)`` = spmm(adj, input @ weight) + bias) on the fused GCN layer op (autograd.gcn_layer)."""
import math

import torch
from torch.nn import Module
from torch.nn.parameter import Parameter

from ....autograd import gcn_layer
from ....graph import CSRGraph, as_graph


class conGraphConvolutionlayer(Module):

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj):
        graph = adj if isinstance(adj, CSRGraph) else as_graph(adj, input.device)
        return gcn_layer(input, self.weight, graph, self.bias, False)

    def __repr__(self):
        return self.__class__.__name__ + " (" + str(self.in_features) + " -> " + str(self.out_features) + ")"
