"""Oracle for the GCN layer family (test infrastructure; see oracle/__init__.py).

Restates, with torch-CPU ops (the same ops the reference executes on its CPU path):
  * GNNLayer          dance/modules/single_modality/clustering/scdsc.py:475-501
  * GraphConvolution  dance/modules/spatial/spatial_domain/spagcn.py:337-366
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


class GNNLayer(nn.Module):
    """scdsc.py:487-501 — weight [in,out] xavier_uniform, no bias; relu(spmm(adj, mm(x, W)))."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))  # scdsc.py:491
        nn.init.xavier_uniform_(self.weight)  # scdsc.py:492

    def forward(self, features, adj, active=True):
        support = torch.mm(features, self.weight)  # scdsc.py:497
        output = torch.spmm(adj, support)  # scdsc.py:498
        if active:
            output = F.relu(output)  # scdsc.py:499-500
        return output


class GraphConvolution(nn.Module):
    """spagcn.py:340-363 — weight/bias ~ U(+-1/sqrt(out)); spmm(adj, mm(x, W)) + b, adj may be dense."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(in_features, out_features))  # spagcn.py:344
        if bias:
            self.bias = nn.Parameter(torch.empty(out_features))  # spagcn.py:346
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(1))  # spagcn.py:352
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj):
        support = torch.mm(input, self.weight)  # spagcn.py:358
        output = torch.spmm(adj, support)  # spagcn.py:359
        if self.bias is not None:
            return output + self.bias  # spagcn.py:360-361
        return output


def scipy_to_torch_coo(mat) -> torch.Tensor:
    """dance/transforms/preprocess.py:526-532 — scipy -> f32 COO torch sparse tensor (int64 indices)."""
    coo = mat.tocoo().astype(np.float32)
    idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data), torch.Size(coo.shape))


def gcn_layer_fwd_bwd(x: np.ndarray, adj, weight: np.ndarray, dy: np.ndarray, *, bias=None, active=True,
                      x_requires_grad=False):
    """One forward+backward of out = act(A (x W) + b) on torch-CPU; returns dict of numpy arrays.

    ``adj`` is a scipy sparse matrix (-> COO as the reference does) or a dense ndarray (SpaGCN).
    """
    xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).requires_grad_(x_requires_grad)
    wt = torch.from_numpy(np.ascontiguousarray(weight, dtype=np.float32)).requires_grad_(True)
    bt = None if bias is None else torch.from_numpy(np.asarray(bias, dtype=np.float32)).requires_grad_(True)
    at = torch.from_numpy(np.asarray(adj, dtype=np.float32)) if isinstance(adj, np.ndarray) else scipy_to_torch_coo(adj)
    out = torch.spmm(at, torch.mm(xt, wt))
    if bt is not None:
        out = out + bt
    if active:
        out = F.relu(out)
    out.backward(torch.from_numpy(np.ascontiguousarray(dy, dtype=np.float32)))
    res = {"out": out.detach().numpy(), "dW": wt.grad.numpy()}
    if x_requires_grad:
        res["dX"] = xt.grad.numpy()
    if bt is not None:
        res["db"] = bt.grad.numpy()
    return res
