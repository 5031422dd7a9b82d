"""STAGATE on MI355X — drop-in for dance/modules/spatial/spatial_domain/stagate.py:31-330 (``GATConv`` :31-128,
``Stagate`` :131-330): same constructors, parameter names (``conv{1..4}.lin_src`` / ``att_src`` / ``att_dst``) and
``forward(features, edge_index)`` / ``fit((x, edge_index_array), ...)`` / ``predict`` signatures.

The reference's GATConv is a torch_geometric ``MessagePassing`` layer: it materialises x_j [E, 1, C], the per-edge logits,
a scatter-softmax and a scatter-add.  Here a layer is: one MFMA GEMM (x @ lin_src), two mat-vecs for the per-node attention
terms, dh_edge_softmax_f32 (sigmoid + softmax over each destination's in-edges, one wavefront per row) and the CSR SpMM with
the attention as edge values — with the matching hand-written backward (autograd.gat_aggregate).  ``edge_index`` (PyG
convention: row 0 = source j, row 1 = target i) is converted once per tensor to a device CSR keyed by destination; the
per-edge attention is returned in the ORIGINAL edge order when ``return_attention_weights`` is used.
"""
import logging
from typing import Any, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor
from torch.nn import Parameter

from .... import kernels
from ....autograd import gat_aggregate
from ....graph import CSRGraph, TensorKeyedCache
from ....transforms import Compose, SetConfig
from ....transforms.graph import StagateGraph
from ...base import BaseClusteringMethod, BasePretrain

logger = logging.getLogger("dance")

_EDGE_CACHE = TensorKeyedCache()
_LOOP_CACHE = TensorKeyedCache()


def edge_index_graph(edge_index: torch.Tensor, n: int):
    """CSR by destination of a PyG ``edge_index`` [2, E] (duplicates kept) + ``slot_of_edge`` (position of every original
    edge in the CSR arrays).  Cached per tensor: the reference passes the same ``edge_index`` to all four layers, every epoch."""
    hit = _EDGE_CACHE.get(edge_index, n)  # identity + in-place version: never a recycled allocation or an edited tensor
    if hit is None:
        src, dst = edge_index[0].long(), edge_index[1].long()
        order = torch.argsort(dst * n + src, stable=True)
        rowptr = torch.zeros(n + 1, dtype=torch.int64, device=edge_index.device)
        rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0)
        g = CSRGraph(rowptr.to(torch.int32), src[order].to(torch.int32).contiguous(), None, n, n)
        slot = torch.empty_like(order)
        slot[order] = torch.arange(order.numel(), device=order.device)
        hit = _EDGE_CACHE.put(edge_index, (g, slot), n)
    return hit


class GATConv(nn.Module):
    """Graph attention layer of STAGATE: sigmoid attention logits, softmax over the in-edges, shared (tied) projection."""
    _alpha = None

    def __init__(self, in_channels, out_channels, heads: int = 1, concat: bool = True, negative_slope: float = 0.2,
                 dropout: float = 0.0, add_self_loops=True, bias=True, **kwargs):
        super().__init__()
        if heads != 1:
            raise NotImplementedError("STAGATE uses single-head attention (stagate.py:147-150); heads > 1 is not implemented on HIP")
        self.in_channels, self.out_channels, self.heads, self.concat = in_channels, out_channels, heads, concat
        self.negative_slope, self.dropout, self.add_self_loops = negative_slope, dropout, add_self_loops
        self.lin_src = nn.Parameter(torch.zeros(size=(in_channels, out_channels)))
        nn.init.xavier_normal_(self.lin_src.data, gain=1.414)
        self.lin_dst = self.lin_src
        self.att_src = Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = Parameter(torch.empty(1, heads, out_channels))
        nn.init.xavier_normal_(self.att_src.data, gain=1.414)
        nn.init.xavier_normal_(self.att_dst.data, gain=1.414)
        self._alpha = None
        self.attentions = None

    def forward(self, x, edge_index, size=None, return_attention_weights=None, attention=True, tied_attention=None):
        if not isinstance(x, Tensor):
            raise NotImplementedError("bipartite (x_src, x_dst) inputs are not used by STAGATE")
        assert x.dim() == 2, "Static graphs not supported in GATConv"
        xs = _mm(x, self.lin_src)  # torch.mm(x, self.lin_src) on the matrix cores
        if not attention:
            return xs  # x[0].mean(dim=1) over the single head
        if tied_attention is None:
            alpha = ((xs * self.att_src.reshape(1, -1)).sum(-1), (xs * self.att_dst.reshape(1, -1)).sum(-1))
            self.attentions = (alpha[0][:, None], alpha[1][:, None])  # [N, heads], as the reference stores them
        else:
            alpha = (tied_attention[0].reshape(-1), tied_attention[1].reshape(-1))
        n = xs.shape[0]
        if self.add_self_loops:  # the augmented list is derived once per caller tensor, so its CSR below is cached too
            aug = _LOOP_CACHE.get(edge_index, n)
            if aug is None:
                keep = edge_index[0] != edge_index[1]
                loops = torch.arange(n, device=edge_index.device)
                aug = _LOOP_CACHE.put(edge_index, torch.cat((edge_index[:, keep], torch.stack((loops, loops))), dim=1), n)
            edge_index = aug
        graph, slot = edge_index_graph(edge_index, n)
        out, att = gat_aggregate(xs, alpha[0], alpha[1], graph, act=kernels.ATT_SIGMOID)
        if self.dropout and self.training:
            raise NotImplementedError("attention dropout > 0 is not used by STAGATE (dropout=0, stagate.py:147-150)")
        if isinstance(return_attention_weights, bool):
            return out, (edge_index, att[slot][:, None])
        return out

    def __repr__(self):
        return "{}({}, {}, heads={})".format(self.__class__.__name__, self.in_channels, self.out_channels, self.heads)


def _mm(x, w):
    from ....autograd import linear
    return linear(x, w.t())  # x @ w through the MFMA GEMM with autograd (linear computes x W'^T)


class Stagate(nn.Module, BasePretrain, BaseClusteringMethod):

    def __init__(self, hidden_dims, device: str = "auto", pretrain_path: Optional[str] = None):
        super().__init__()
        self.pretrain_path = pretrain_path
        [in_dim, num_hidden, out_dim] = hidden_dims
        self.conv1 = GATConv(in_dim, num_hidden, heads=1, concat=False, dropout=0, add_self_loops=False, bias=False)
        self.conv2 = GATConv(num_hidden, out_dim, heads=1, concat=False, dropout=0, add_self_loops=False, bias=False)
        self.conv3 = GATConv(out_dim, num_hidden, heads=1, concat=False, dropout=0, add_self_loops=False, bias=False)
        self.conv4 = GATConv(num_hidden, in_dim, heads=1, concat=False, dropout=0, add_self_loops=False, bias=False)
        self.device = "cuda" if device == "auto" else device
        self.to(self.device)

    @staticmethod
    def preprocessing_pipeline(hvg_flavor: str = "seurat_v3", n_top_hvgs: int = 3000, model_name: str = "radius", radius: float = 150,
                               n_neighbors: int = 5, log_level="INFO"):
        """stagate.py:157-173: HVG selection, normalize_total to 1e4 and log1p on the device, then the spatial graph.  The
        reference's default "seurat_v3" (a loess fit of the count variances) and the dispersion flavours ("seurat",
        "cell_ranger") all run on the device; ``hvg_flavor=None`` skips the selection."""
        from ....transforms import HighlyVariableGenesLogarithmizedByTopGenes, HighlyVariableGenesRawCount, Log1P, NormalizeTotal
        steps = []
        if hvg_flavor in ("seurat", "cell_ranger"):
            steps.append(HighlyVariableGenesLogarithmizedByTopGenes(n_top_genes=n_top_hvgs, flavor=hvg_flavor, subset=True))
        elif hvg_flavor == "seurat_v3":
            steps.append(HighlyVariableGenesRawCount(n_top_genes=n_top_hvgs, subset=True))
        elif hvg_flavor is not None:
            raise ValueError(f"unknown hvg_flavor {hvg_flavor!r}")
        return Compose(
            *steps,
            NormalizeTotal(target_sum=1e4, max_fraction=1.0),
            Log1P(),
            StagateGraph(model_name, radius=radius, n_neighbors=n_neighbors),
            SetConfig({"feature_channel": "StagateGraph", "feature_channel_type": "obsp", "label_channel": "label", "label_channel_type": "obs"}),
            log_level=log_level,
        )

    def forward(self, features, edge_index):
        h1 = F.elu(self.conv1(features, edge_index))
        h2 = self.conv2(h1, edge_index, attention=False)
        self.conv3.lin_src.data = self.conv2.lin_src.transpose(0, 1)
        self.conv3.lin_dst.data = self.conv2.lin_dst.transpose(0, 1)
        self.conv4.lin_src.data = self.conv1.lin_src.transpose(0, 1)
        self.conv4.lin_dst.data = self.conv1.lin_dst.transpose(0, 1)
        h3 = F.elu(self.conv3(h2, edge_index, attention=True, tied_attention=self.conv1.attentions))
        h4 = self.conv4(h3, edge_index, attention=False)
        return h2, h4

    def pretrain(self, x: np.ndarray, edge_index_array: np.ndarray, lr: float = 1e-3, weight_decay: float = 1e-4, epochs: int = 100,
                 gradient_clipping: float = 5):
        x_tensor = torch.from_numpy(x.astype(np.float32)).to(self.device)
        edge_index_tensor = torch.from_numpy(edge_index_array.astype(int)).to(self.device)
        # foreach=False: after the first forward conv3 / conv4's weights are transposed VIEWS of conv2 / conv1's storage
        # (:191-194), so the order in which Adam touches the aliased memory matters; the per-parameter loop is what the
        # reference's CPU run executes (the multi-tensor path would read all old values first)
        optimizer = torch.optim.Adam(self.parameters(), lr=lr, weight_decay=weight_decay, foreach=False)
        self.train()
        for _ in range(1, epochs + 1):
            optimizer.zero_grad()
            z, out = self(x_tensor, edge_index_tensor)
            loss = F.mse_loss(x_tensor, out)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(self.parameters(), gradient_clipping)
            optimizer.step()
        self.eval()
        z, _ = self(x_tensor, edge_index_tensor)
        self.rep = z.detach().clone().cpu().numpy()

    def save_pretrained(self, path):
        np.save(path, self.rep)

    def load_pretrained(self, path):
        self.rep = np.load(path)

    def fit(self, inputs: Tuple[np.ndarray, np.ndarray], epochs: int = 100, lr: float = 0.001, gradient_clipping: float = 5,
            weight_decay: float = 1e-4, num_cluster: int = 7, gmm_reg_covar: float = 1.5e-4, gmm_n_init: int = 10, gmm_max_iter: int = 300,
            gmm_tol: float = 2e-4, random_state: Optional[int] = None):
        from sklearn.mixture import GaussianMixture
        x, edge_index_array = inputs
        self._pretrain(x, edge_index_array, lr, weight_decay, epochs, gradient_clipping)
        gmm = GaussianMixture(n_components=num_cluster, covariance_type="tied", n_init=gmm_n_init, tol=gmm_tol, max_iter=gmm_max_iter,
                              reg_covar=gmm_reg_covar, random_state=random_state)
        self.clust_res = gmm.fit_predict(self.rep)

    def predict(self, x: Optional[Any] = None):
        return self.clust_res
