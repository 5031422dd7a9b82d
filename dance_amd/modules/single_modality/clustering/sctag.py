"""scTAG on MI355X — drop-in for dance/modules/single_modality/clustering/sctag.py:32-548 (``ScTAG`` :32-430,
``DecoderAdj`` :433-472, ``DecoderX`` :475-528) with ``TAGConv`` (dgl.nn.TAGConv, which the reference imports at :19)
implemented on the CSR SpMM.

``TAGConv`` (K-hop topology-adaptive convolution, dgl 1.1.3 [3P-memory]): the hop recursion h_{k+1} = A_hat h_k with
  * no edge weights:  A_hat = D_in^-1/2 A D_in^-1/2, D_in = in-degrees clamped at 1 (applied to source and destination side);
  * edge weights w:   A_hat[v, u] = w_uv / sqrt(sum_out_w(u) * sum_in_w(v))  (``EdgeWeightNorm("both")``)
is K launches of dh_spmm_csr_f32 with the two scale vectors fused as rowscale / colscale (no [E, D] messages, no scaled copies
of h), followed by ONE Linear over [h_0 | h_1 | ... | h_K] on the matrix cores.  Backward = the same kernel on the cached
transposed CSR.

``DecoderAdj`` is the reference's dense decoder (a Linear latent -> N, then an N x N inner product, :455-472): it is what the
reference computes and is kept for parity at the sizes where an N x N matrix exists at all (the GEMMs run on dh_gemm_f32).
"""
from typing import Any, Optional, Tuple

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from torch.nn import Parameter

from .... import kernels
from ....autograd import HipLinear, adj_reconstruction_mse, linear, spmm
from ....graph import CSRGraph
from ....transforms import CellPCA, Compose, SetConfig
from ....transforms.graph import NeighborGraph
from ...base import BaseClusteringMethod, TorchNNPretrain
from .scdsc import DispAct, MeanAct, ZINBLoss


def dist_loss(data, min_dist, max_dist=20):
    """dance/utils/loss.py:832-842."""
    pairwise_dist = (torch.sum((data.unsqueeze(1) - data.unsqueeze(0))**2, 2) + 1e-10)**(1 / 2)
    return torch.exp(-(pairwise_dist - min_dist)) + torch.exp(-(max_dist - pairwise_dist))


class WeightedGraph:
    """What the reference keeps as ``dgl.graph((src, dst))`` + ``edata["weight"]`` (:95-97): a device CSR by DESTINATION
    (row v lists the sources u of its in-edges, ascending) with the edge weights, plus the TAGConv degree factor."""

    def __init__(self, adj, device):
        a = sp.csr_matrix(adj, dtype=np.float32) if sp.issparse(adj) else sp.csr_matrix(np.asarray(adj, dtype=np.float32))
        a.eliminate_zeros()  # np.nonzero(adj)
        at = a.T.tocsr()     # edge (src = row, dst = col): aggregate at the destination
        at.sort_indices()
        self.csr = CSRGraph.from_scipy(at, device)
        self.edata = {"weight": self.csr.val}
        in_deg = torch.from_numpy(np.diff(at.indptr).astype(np.float32)).to(device)
        self.norm = in_deg.clamp(min=1).pow(-0.5)
        self.unweighted = CSRGraph(self.csr.rowptr, self.csr.col, None, self.csr.n_rows, self.csr.n_cols)
        self._ew_scales = {}

    def edge_weight_norm(self, w: torch.Tensor):
        """dgl.nn.EdgeWeightNorm("both") as (rowscale over destinations, colscale over sources): weighted in / out sums ^-1/2."""
        key = w.data_ptr()
        if key not in self._ew_scales:
            if bool((w <= 0).any()):
                raise RuntimeError("Non-positive edge weight detected with `norm=\"both\"`. This leads to square root of zero or negative values.")
            n = self.csr.n_rows
            rows = torch.repeat_interleave(torch.arange(n, device=w.device), (self.csr.rowptr[1:] - self.csr.rowptr[:-1]).long(), output_size=w.numel())
            in_sum = torch.zeros(n, device=w.device).index_add_(0, rows, w)
            out_sum = torch.zeros(self.csr.n_cols, device=w.device).index_add_(0, self.csr.col.long(), w)
            self._ew_scales = {key: (in_sum.pow(-0.5), out_sum.pow(-0.5))}
        return self._ew_scales[key]

    def number_of_nodes(self):
        return self.csr.n_rows


class TAGConv(nn.Module):
    """dgl.nn.TAGConv(in_feats, out_feats, k=2, bias=True, activation=None): parameters ``lin.weight`` [out, in * (k + 1)],
    ``lin.bias``; xavier-normal init with the ReLU gain, as DGL's ``reset_parameters``."""

    def __init__(self, in_feats, out_feats, k=2, bias=True, activation=None):
        super().__init__()
        self._in_feats, self._out_feats, self._k, self._activation = in_feats, out_feats, k, activation
        self.lin = HipLinear(in_feats * (k + 1), out_feats, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_normal_(self.lin.weight, gain=nn.init.calculate_gain("relu"))

    def forward(self, graph: WeightedGraph, feat, edge_weight=None):
        if edge_weight is None:
            g, rs, cs = graph.unweighted, graph.norm, graph.norm
        else:
            w = edge_weight.reshape(-1).contiguous()
            g = graph.csr if w.data_ptr() == graph.csr.val.data_ptr() else CSRGraph(graph.csr.rowptr, graph.csr.col, w, graph.csr.n_rows, graph.csr.n_cols)
            rs, cs = graph.edge_weight_norm(w)
        fstack = [feat]
        for _ in range(self._k):
            fstack.append(spmm(fstack[-1], g, rowscale=rs, colscale=cs))
        rst = self.lin(torch.cat(fstack, dim=-1))
        return rst if self._activation is None else self._activation(rst)


class DecoderAdj(nn.Module):

    def __init__(self, latent_dim=15, adj_dim=32, activation=torch.sigmoid, dropout=0):
        super().__init__()
        self.dropout = dropout
        self.activation = activation
        self.dec_1 = HipLinear(latent_dim, adj_dim)

    def forward(self, z, factor_only: bool = False):
        z0 = F.dropout(self.dec_1(z), self.dropout)  # training=True always, as in the reference (:470)
        if factor_only:  # the scalable loss works on the factor: sigmoid(z0 z0^T) is never formed (autograd.adj_reconstruction_mse)
            return z0
        return self.activation(linear(z0, z0))        # z0 z0^T on the matrix cores


class DecoderX(nn.Module):

    def __init__(self, input_dim, n_z, n_dec_1=128, n_dec_2=256, n_dec_3=512):
        super().__init__()
        self.n_dec_3 = n_dec_3
        self.input_dim = input_dim
        self.dec_1 = HipLinear(n_z, n_dec_1)
        self.dec_2 = HipLinear(n_dec_1, n_dec_2)
        self.dec_3 = HipLinear(n_dec_2, n_dec_3)
        self.dec_mean = nn.Sequential(HipLinear(self.n_dec_3, self.input_dim), MeanAct())
        self.dec_disp = nn.Sequential(HipLinear(self.n_dec_3, self.input_dim), DispAct())
        self.dec_pi = nn.Sequential(HipLinear(self.n_dec_3, self.input_dim), nn.Sigmoid())

    def forward(self, z):
        return self._forward(z, False)

    def _forward(self, z, raw_heads: bool):
        """``raw_heads=True`` (the two training loops): the heads' Linear outputs, for ``ZINBLoss.from_logits`` — MeanAct / DispAct /
        Sigmoid then run inside the loss kernels (see scdsc.ScDSCModel._forward)."""
        dec_h3 = self.dec_3(self.dec_2(self.dec_1(z, fuse_relu=True), fuse_relu=True), fuse_relu=True)
        if raw_heads:
            return self.dec_mean[0](dec_h3), self.dec_disp[0](dec_h3), self.dec_pi[0](dec_h3)
        return self.dec_mean(dec_h3), self.dec_disp(dec_h3), self.dec_pi(dec_h3)


class ScTAG(nn.Module, TorchNNPretrain, BaseClusteringMethod):

    def __init__(self, n_clusters: int, k: int = 3, hidden_dim: int = 128, latent_dim: int = 15, dec_dim: Optional[int] = None,
                 dropout: float = 0.2, device: str = "cuda", alpha: float = 1.0, pretrain_path: Optional[str] = None, *,
                 adj_dim: Optional[int] = None):
        super().__init__()
        # adj_dim=None: the reference's adjacency decoder — a Linear(latent_dim -> N) followed by an N x N x N product and an N x N
        # target (sctag.py:302, :433-472): parameters, memory and flops grow with the number of cells.  adj_dim=d (<= 320): the same
        # decoder with a width that does not depend on N, and the loss mean((sigmoid(z0 z0^T) - adj)^2) over ALL N^2 pairs evaluated
        # without any N x N matrix (dh_gram_pairwise_f32 + dh_sddmm_csr_f32, autograd.adj_reconstruction_mse): O(N d) memory.
        self.adj_dim = adj_dim
        self._is_pretrained = False
        self._in_dim = None
        self.pretrain_path = pretrain_path
        self.dec_dim = dec_dim or [128, 256, 512]
        self.latent_dim = latent_dim
        self.hidden_dim = hidden_dim
        self.device = "cuda" if device == "auto" else device
        self.dropout = dropout
        self.n_clusters = n_clusters
        self.alpha = alpha
        self.k = k

    def init_model(self, adj, x: np.ndarray):
        self._in_dim = x.shape[1]
        if self.adj_dim is not None:  # scalable mode: the adjacency stays sparse from here on
            a = sp.csr_matrix(adj, dtype=np.float32)
            deg = np.asarray(a.sum(1)).ravel()
            deg[deg == 0] = 1
            dis = sp.diags((deg**-0.5).astype(np.float32))
            adj_n = (dis @ a @ dis).tocsr()
            self.g = WeightedGraph((a != 0).astype(np.float32), self.device)
            self.g_n = WeightedGraph(adj_n, self.device)
            a.sort_indices()
            self.adj_target = CSRGraph.from_scipy(a, self.device, symmetric=(abs(a - a.T) > 0).nnz == 0)
        else:
            adj = np.asarray(adj, dtype=np.float32)
            deg = adj.sum(1, keepdims=True)
            deg[deg == 0] = 1
            normalized_deg = deg**-0.5
            adj_n = adj * normalized_deg * normalized_deg.T
            self.g = WeightedGraph((adj != 0).astype(np.float32), self.device)
            self.g_n = WeightedGraph(adj_n, self.device)
            self.adj_target = None
        self.mu = Parameter(torch.empty(self.n_clusters, self.latent_dim, device=self.device))
        self.encoder1 = TAGConv(self.in_dim, self.hidden_dim, k=self.k)
        self.encoder2 = TAGConv(self.hidden_dim, self.latent_dim, k=self.k)
        self.decoder_adj = DecoderAdj(latent_dim=self.latent_dim, adj_dim=self.adj_dim or adj.shape[0], activation=torch.sigmoid,
                                      dropout=self.dropout)
        self.decoder_x = DecoderX(self.in_dim, self.latent_dim, n_dec_1=self.dec_dim[0], n_dec_2=self.dec_dim[1], n_dec_3=self.dec_dim[2])
        self.zinb_loss = ZINBLoss().to(self.device)
        self.to(self.device)

    @property
    def in_dim(self) -> int:
        if self._in_dim is None:
            raise ValueError("in_dim is unavailable since the model has not been initialized yet. Please call the "
                             "`fit` function first to fit the model, or the `init_model` function "
                             "if you just want to initialize the model.")
        return self._in_dim

    @staticmethod
    def preprocessing_pipeline(n_top_genes: int = 3000, n_components: int = 50, n_neighbors: int = 15, log_level="INFO"):
        """sctag.py:119-145, every step on the device (DeviceArray slots, no host round trip between steps): filter genes / cells,
        per-cell normalisation to the median count (``sc.pp.normalize_per_cell``, ``n_counts`` recorded), log1p, cell_ranger HVG, second
        filter, SaveRaw, normalize_total, log1p, scale, then PCA (on the device) and the kNN graph in PCA space."""
        from ....transforms import (FilterCellsScanpy, FilterGenesScanpy, HighlyVariableGenesLogarithmizedByTopGenes, Log1P, NormalizeTotal,
                                    SaveRaw, Scale)
        return Compose(
            FilterGenesScanpy(min_counts=3),
            FilterCellsScanpy(min_counts=1, key_n_counts="n_counts"),
            NormalizeTotal(max_fraction=1.0, key_added="n_counts"),
            Log1P(),
            HighlyVariableGenesLogarithmizedByTopGenes(n_top_genes=n_top_genes, flavor="cell_ranger", subset=True),
            FilterGenesScanpy(min_counts=1),
            FilterCellsScanpy(min_counts=1, key_n_counts="n_counts"),
            SaveRaw(),
            NormalizeTotal(max_fraction=1.0),
            Log1P(),
            Scale(),
            CellPCA(n_components=n_components, device="cuda"),
            NeighborGraph(n_neighbors=n_neighbors, n_pcs=n_components),
            SetConfig({"feature_channel": ["NeighborGraph", None, None, "n_counts"],
                       "feature_channel_type": ["obsp", "X", "raw_X", "obs"], "label_channel": "Group"}),
            log_level=log_level,
        )

    def forward(self, g, x_input):
        return self._forward(g, x_input, False)

    def _forward(self, g, x_input, raw_heads: bool):
        enc_h = self.encoder1(g, x_input, edge_weight=g.edata["weight"])
        z = self.encoder2(g, enc_h, edge_weight=g.edata["weight"])
        adj_out = self.decoder_adj(z, factor_only=self.adj_dim is not None)  # scalable mode: the factor z0 [N, adj_dim], not N x N
        _mean, _disp, _pi = self.decoder_x._forward(z, raw_heads)
        return adj_out, z, self.soft_assign(z), _mean, _disp, _pi

    def adj_loss(self, adj_out, adj_t):
        """``torch.mean(F.mse_loss(adj_out, adj))`` of sctag.py:254 / :347; in scalable mode from the decoder's factor and the sparse
        target, over the same N^2 entries."""
        if self.adj_dim is not None:
            return adj_reconstruction_mse(adj_out, self.adj_target)
        return torch.mean(F.mse_loss(adj_out, adj_t))

    def pretrain(self, adj, x, x_raw, n_counts, *, epochs: int = 1000, info_step: int = 10, lr: float = 5e-4, w_a: float = 0.3,
                 w_x: float = 1, w_d: float = 0, min_dist: float = 0.5, max_dist: float = 20, force_pretrain: bool = False):
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self.device)
        x_raw = torch.as_tensor(np.asarray(x_raw), dtype=torch.float32).to(self.device)
        n_counts = np.asarray(n_counts, dtype=np.float64)
        scale_factor = torch.as_tensor(n_counts / np.median(n_counts)).to(self.device)
        adj_t = None if self.adj_dim is not None else torch.as_tensor(np.asarray(adj), dtype=torch.float32).to(self.device)
        self.train()
        optimizer = optim.Adam(filter(lambda p: p.requires_grad, self.parameters()), lr=lr, amsgrad=True)
        for _ in range(epochs):
            adj_out, z, _, mean, disp, pi = self._forward(self.g_n, x, True)  # raw head outputs: the activations run inside the loss kernels
            loss = w_a * self.adj_loss(adj_out, adj_t) + w_x * self.zinb_loss.from_logits(x_raw, mean, disp, pi, scale_factor)
            if w_d:
                loss = loss + w_d * torch.mean(dist_loss(z, min_dist, max_dist=max_dist))
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()

    def save_pretrained(self, path):
        torch.save(self.state_dict(), path)

    def load_pretrained(self, path):
        self.load_state_dict(torch.load(path, map_location=self.device))

    def fit(self, inputs: Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray], y: np.ndarray, *, epochs: int = 300,
            pretrain_epochs: int = 200, lr: float = 5e-4, w_a: float = 0.3, w_x: float = 1, w_c: float = 1.5, w_d: float = 0,
            info_step: int = 1, max_dist: float = 20, min_dist: float = 0.5, force_pretrain: bool = False):
        from sklearn.cluster import KMeans
        adj, x, x_raw, n_counts = inputs
        if hasattr(adj, "materialize"):  # LazyScipyCSR left by an on-device NeighborGraph
            adj = adj.materialize()
        if sp.issparse(adj) and self.adj_dim is None:
            adj = np.asarray(adj.todense())
        self.init_model(adj, x)
        self._pretrain(adj, x, x_raw, n_counts, epochs=pretrain_epochs, info_step=info_step, lr=lr, w_a=w_a, w_x=w_x, w_d=w_d,
                       min_dist=min_dist, max_dist=max_dist, force_pretrain=True)
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32).to(self.device)
        x_raw = torch.as_tensor(np.asarray(x_raw), dtype=torch.float32).to(self.device)
        n_counts = np.asarray(n_counts, dtype=np.float64)
        scale_factor = torch.as_tensor(n_counts / np.median(n_counts)).to(self.device)
        adj_t = None if self.adj_dim is not None else torch.as_tensor(np.asarray(adj), dtype=torch.float32).to(self.device)
        # cluster centres from kmeans on the embedding — computed WITHOUT edge weights, as the reference does (:313-314)
        kmeans = KMeans(self.n_clusters, n_init=20)
        z = self.encoder2(self.g_n, self.encoder1(self.g_n, x))
        kmeans.fit_predict(z.detach().cpu().numpy())
        self.mu.data.copy_(torch.tensor(kmeans.cluster_centers_, dtype=torch.float32).to(self.device))
        self.train()
        optimizer = optim.Adam(filter(lambda p: p.requires_grad, self.parameters()), lr=lr, amsgrad=True)
        aris, Q = [], {}
        y_t = torch.as_tensor(np.asarray(y), dtype=torch.float32).to(self.device)
        for epoch in range(epochs):
            adj_out, _, q, mean, disp, pi = self._forward(self.g_n, x, True)
            self.q = q
            self.y_pred = self.predict()
            aris.append(self.score(None, y))  # ARI for model selection (:331-333)
            Q[f"epoch{epoch}"] = q
            # the "cluster loss" of :343-346 is a KL between two constant label vectors: it carries no gradient, but its value
            # (possibly nan / inf) is added to the loss exactly as the reference does
            cluster_loss = torch.mean(F.kl_div(torch.as_tensor(self.y_pred, dtype=torch.float32).to(self.device), y_t, reduction="batchmean"))
            loss = w_a * self.adj_loss(adj_out, adj_t) + w_x * self.zinb_loss.from_logits(x_raw, mean, disp, pi, scale_factor) + w_c * cluster_loss
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        self.q = Q[f"epoch{int(np.argmax(aris))}"]

    def predict_proba(self, x: Optional[Any] = None) -> np.ndarray:
        return self.q.detach().clone().cpu().numpy()

    def predict(self, x: Optional[Any] = None) -> np.ndarray:
        return self.predict_proba().argmax(1)

    def soft_assign(self, z):
        q = 1.0 / (1.0 + torch.sum((z.unsqueeze(1) - self.mu)**2, dim=2) / self.alpha)
        q = q**((self.alpha + 1.0) / 2.0)
        return (q.t() / torch.sum(q, dim=1)).t()

    def target_distribution(self, q):
        p = q**2 / q.sum(0)
        return (p.t() / p.sum(1)).t()
