"""scDSC's GCN layer on MI355X — drop-in for ``GNNLayer`` of
dance/modules/single_modality/clustering/scdsc.py:475-501 (same constructor, parameter name ``weight``,
init and ``forward(features, adj, active=True)`` signature, so reference checkpoints load unchanged).

``adj`` may be the torch sparse tensor the reference passes (scdsc.py:244) or a ``dance_amd.graph.CSRGraph``.
"""
import torch
from torch import nn

from ....autograd import gcn_layer
from ....graph import as_graph


class GNNLayer(nn.Module):

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        torch.nn.init.xavier_uniform_(self.weight)

    def forward(self, features, adj, active=True):
        # relu(spmm(adj, mm(features, weight))): MFMA GEMM + fused-ReLU CSR SpMM (dance_amd/autograd.py)
        return gcn_layer(features, self.weight, as_graph(adj, features.device), None, bool(active))
