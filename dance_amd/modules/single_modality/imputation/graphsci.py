"""GraphSCI on MI355X — drop-in for dance/modules/single_modality/imputation/graphsci.py.  The gene-graph network ``GNNModel``
(:107-124): three ``dgl.nn.GraphConv`` layers (norm="both", tanh / relu) and the
two decoders over the gene-gene graph.  ``GraphConv`` here is DGL's layer (parameters ``weight`` [in, out] xavier-uniform,
``bias`` zeros; out = D_in^-1/2 A D_out^-1/2 X W + b with degrees clamped at 1) as ONE fused GCN layer op: the two degree
factors ride the SpMM as row / column scales (autograd.gcn_layer), the activation's ReLU in its epilogue.

``forward(g)``: ``g`` is a ``CSRGraph`` (rows = destination genes) carrying ``g.ndata["feat"]``, which is what the
FeatureFeatureGraph of the reference holds in its DGL graph.  The reference's quirk of computing ``z_adj_log_std`` with
``dec_mean`` (:121) is kept.

``AEModel`` (:36-104) is the expression auto-encoder: ``MultiplyLayer`` = X (adj W^T) + b as two chained products of the exact-fp32
GEMM (``dense_adj_layer``), the encoder / three decoders as ``Linear -> BatchNorm1d -> activation`` stacks on ``HipLinear``
(state_dict keys as the reference's ``nn.Sequential``).  ``GraphSCI`` (:126-560) ties them together; where the reference pulls
every tensor to the CPU to evaluate its loss (:466-471), the loss here stays on the device.  There is no CPU path: ``gpu=-1`` means
the current device."""
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ....autograd import HipLinear, dense_adj_layer, gcn_layer
from ....graph import CSRGraph
from ...base import BaseRegressionMethod


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


def buildNetwork(layers, dropout=0., activation=nn.ReLU()):
    net = []
    for i in range(1, len(layers)):
        net += [nn.Dropout(dropout), HipLinear(layers[i - 1], layers[i]), nn.BatchNorm1d(layers[i]), activation]
    return nn.Sequential(*net)


class DispActivation(nn.Module):

    def forward(self, input):
        return torch.clamp(F.softplus(input), 1e-4, 1e4)


class MeanActivation(nn.Module):

    def forward(self, input):
        return torch.clamp(torch.exp(input), 1e-5, 1e6)


class MultiplyLayer(nn.Module):
    """act(dropout(X) @ fc(adj) + bias), fc a bias-free G x G linear map of the (generated) adjacency (graphsci.py:64-84)."""

    def __init__(self, num_nodes, dropout=0., act=nn.ReLU(), bias=True):
        super().__init__()
        self.num_nodes, self.dropout, self.act = num_nodes, dropout, act
        self.fc_layer = nn.Linear(num_nodes, num_nodes, bias=False)
        self.dp = nn.Dropout(dropout)
        self.bias_flag = bias
        if bias:
            self.bias = nn.Parameter(torch.zeros(num_nodes))

    def forward(self, X, adj):
        z = dense_adj_layer(adj, self.fc_layer.weight.t(), self.dp(X), self.bias if self.bias_flag else None)
        return self.act(z)


class AEModel(nn.Module):

    def __init__(self, in_feats, dropout=0., n_hidden1=256, n_hidden2=256):
        super().__init__()
        self.mul_layer = MultiplyLayer(in_feats, dropout)
        self.enc = buildNetwork([in_feats, n_hidden1, n_hidden2], dropout)
        self.dec_pi = buildNetwork([n_hidden2, in_feats], dropout, nn.Sigmoid())
        self.dec_disp = buildNetwork([n_hidden2, in_feats], dropout, DispActivation())
        self.dec_mean = buildNetwork([n_hidden2, in_feats], dropout, MeanActivation())

    def forward(self, X, adj, size_factors):
        h = self.enc(self.mul_layer(X, adj))
        pi, disp, mean = self.dec_pi(h), self.dec_disp(h), self.dec_mean(h)
        x_exp = mean * torch.reshape(size_factors, (-1, 1))
        return x_exp, mean, disp, pi


def _edges(graph: CSRGraph):
    """(src, dst) of every stored entry: the CSR is by destination (rows), the columns are the sources."""
    counts = (graph.rowptr[1:] - graph.rowptr[:-1]).to(torch.int64)
    dst = torch.repeat_interleave(torch.arange(graph.n_rows, device=graph.rowptr.device), counts)
    return graph.col.to(torch.int64), dst


class GraphSCI(nn.Module, BaseRegressionMethod):
    """GraphSCI: a variational graph auto-encoder over the gene graph generates an adjacency that conditions a ZINB expression
    auto-encoder (graphsci.py:126-560)."""

    def __init__(self, num_cells, num_genes, dataset, dropout=0.1, gpu=-1, seed=1, *, device=None):
        super().__init__()
        self.dataset, self.seed = dataset, seed
        self.prj_path = Path().resolve()
        self.save_path = self.prj_path / "graphsci"
        if not self.save_path.exists():
            self.save_path.mkdir(parents=True)
        self.device = torch.device(device) if device is not None else torch.device("cuda" if gpu == -1 else f"cuda:{gpu}")
        self.gnnmodel = GNNModel(in_feats=num_cells, out_feats=num_genes, dropout=dropout)
        self.aemodel = AEModel(in_feats=num_genes, dropout=dropout)
        self.model_params = list(self.aemodel.parameters()) + list(self.gnnmodel.parameters())
        self.to(self.device)

    @staticmethod
    def preprocessing_pipeline(min_cells: float = 0.1, threshold: float = 0.3, normalize_edges: bool = True, mask: bool = True,
                               distr: str = "exp", mask_rate: float = 0.1, seed: int = 1, log_level="INFO"):
        from ....transforms import (CellwiseMaskData, Compose, FilterCellsScanpy, FilterGenesScanpy, FilterGenesTopK, Log1P, SaveRaw, SetConfig,
                                    UpdateRaw)
        from ....transforms.graph import FeatureFeatureGraph
        transforms = [
            FilterGenesScanpy(min_cells=min_cells),
            FilterCellsScanpy(min_counts=1),
            SaveRaw(),
            Log1P(),
            FilterGenesTopK(num_genes=2000, mode="var"),
            UpdateRaw(),
            FeatureFeatureGraph(threshold=threshold, normalize_edges=normalize_edges),
        ]
        if mask:
            transforms.extend([
                CellwiseMaskData(distr=distr, mask_rate=mask_rate, seed=seed, add_test_mask=True),
                SetConfig({
                    "feature_channel": [None, None, "FeatureFeatureGraph", "train_mask", "valid_mask", "test_mask"],
                    "feature_channel_type": ["X", "raw_X", "uns", "layers", "layers", "layers"],
                    "label_channel": [None, None],
                    "label_channel_type": ["X", "raw_X"],
                })
            ])
        else:
            transforms.append(
                SetConfig({
                    "feature_channel": [None, None, "FeatureFeatureGraph"],
                    "feature_channel_type": ["X", "raw_X", "uns"],
                    "label_channel": [None, None],
                    "label_channel_type": ["X", "raw_X"],
                }))
        return Compose(*transforms, log_level=log_level)

    def maskdata(self, X, mask):
        X_masked = torch.zeros_like(X)
        X_masked[mask] = X[mask]
        return X_masked

    def _mask(self, mask):
        """A boolean N x G mask on the model's device (the reference indexes CPU tensors with numpy masks)."""
        return mask.to(self.device) if isinstance(mask, torch.Tensor) else torch.from_numpy(np.asarray(mask, dtype=bool)).to(self.device)

    def fit(self, train_data, train_data_raw, graph, mask=None, le=1, la=1, ke=1, ka=1, n_epochs=100, lr=1e-3, weight_decay=1e-5,
            train_idx=None):
        n = graph.n_rows
        u, v = _edges(graph)
        self.adj = torch.zeros((n, n), device=self.device)
        self.adj_norm = torch.zeros((n, n), device=self.device)
        self.adj[u, v] = 1.0
        self.adj_norm[u, v] = graph.edata["weight"]
        rng = np.random.default_rng(self.seed)
        n_cells = len(train_data)
        if train_idx is None:
            train_idx = range(n_cells)
        if mask is not None:
            mask = np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask, dtype=bool)
            train_data_masked = self.maskdata(train_data, self._mask(mask))
            graph.ndata["feat"] = train_data_masked.float().T.contiguous()
            test_rows = np.setdiff1d(np.arange(n_cells), np.asarray(list(train_idx), dtype=np.int64))
            train_mask = np.copy(mask)
            train_mask[test_rows] = False
            valid_mask = ~mask
            valid_mask[test_rows] = False
        else:
            train_data_masked = train_data
            permuted = rng.permutation(train_idx)
            cut = int(len(permuted) * 0.9)
            train_mask = np.zeros(tuple(train_data.shape), dtype=bool)
            train_mask[permuted[:cut]] = True
            valid_mask = np.zeros(tuple(train_data.shape), dtype=bool)
            valid_mask[permuted[cut:]] = True
        train_mask, valid_mask = self._mask(train_mask), self._mask(valid_mask)
        self.train_data_masked = train_data_masked
        n_counts = train_data_raw.sum(1)
        self.size_factors = n_counts / torch.median(n_counts)
        self.weight_decay = weight_decay
        self.optimizer = torch.optim.Adam(self.model_params, lr=lr, weight_decay=weight_decay)
        self.save_model()  # a model file exists from the start, as in the reference
        for epoch in range(n_epochs):
            self.train(train_data_masked, train_data_raw, graph, train_mask, valid_mask, le, la, ke, ka)
            if not epoch:
                min_valid_loss = self.valid_loss
            elif min_valid_loss >= self.valid_loss:
                min_valid_loss = self.valid_loss
                self.save_model()
            print(f"[Epoch%d], train_loss %.6f, adj_loss %.6f, express_loss %.6f, kl_loss %.6f, valid_loss %.6f" %
                  (epoch, self.train_loss, self.loss_adj, self.loss_exp, abs(self.kl), self.valid_loss))

    def train(self, train_data, train_data_raw, graph, train_mask, valid_mask, le=1, la=1, ke=1, ka=1):
        """One full-batch step (graphsci.py:283-336); NOTE this shadows ``nn.Module.train`` exactly as the reference does, so the
        sub-modules are switched with their own ``.train()`` / ``.eval()``."""
        self.gnnmodel.train()
        self.aemodel.train()
        self.optimizer.zero_grad()
        z_adj, z_adj_log_std, z_adj_mean = self.gnnmodel.forward(graph)
        z_exp, mean, disp, pi = self.aemodel.forward(train_data, z_adj, self.size_factors)
        loss_adj, loss_exp, log_lik, kl, train_loss = self.get_loss(train_data_raw, self.adj, z_adj, z_adj_log_std, z_adj_mean, z_exp, mean,
                                                                    disp, pi, train_mask, le, la, ke, ka)
        valid_loss, _, _ = self.evaluate(train_data, train_data_raw, graph, valid_mask, le, la, ke, ka)
        self.loss_adj, self.loss_exp, self.log_lik, self.kl = loss_adj.item(), loss_exp.item(), log_lik.item(), kl.item()
        self.train_loss, self.valid_loss = train_loss.item(), valid_loss.item()
        train_loss.backward()
        self.optimizer.step()
        return self.train_loss

    def evaluate(self, features, features_raw, graph, mask=None, le=1, la=1, ke=1, ka=1):
        if mask is None:
            mask = torch.ones(tuple(features_raw.shape), dtype=torch.bool, device=self.device)
        self.aemodel.eval()
        self.gnnmodel.eval()
        with torch.no_grad():
            z_adj, z_adj_log_std, z_adj_mean = self.gnnmodel.forward(graph)
            z_exp, mean, disp, pi = self.aemodel.forward(features, z_adj, self.size_factors)
            _, _, _, _, loss = self.get_loss(features_raw, self.adj, z_adj, z_adj_log_std, z_adj_mean, z_exp, mean, disp, pi, mask, le, la, ke,
                                             ka)
        return loss, z_adj, z_exp

    def save_model(self):
        state = {"aemodel": self.aemodel.state_dict(), "gnnmodel": self.gnnmodel.state_dict(), "optimizer": self.optimizer.state_dict()}
        torch.save(state, self.save_path / f"{self.dataset}.pt")

    def predict(self, data, data_raw, graph, mask=None):
        if mask is not None:
            data = self.maskdata(data, self._mask(mask))
        _, _, z_exp = self.evaluate(data, data_raw, graph)
        return z_exp

    def get_loss(self, batch, adj_orig, z_adj, z_adj_log_std, z_adj_mean, z_exp, mean, disp, pi, mask, le=1, la=1, ke=1, ka=1):
        """(loss_adj, loss_exp, log_lik, kl, loss) as graphsci.py:466-505 computes them — weighted soft-target cross entropy of the
        generated adjacency, masked ZINB negative log-likelihood of the raw counts, the two KL terms — on the device."""
        # reference: dance/modules/single_modality/imputation/graphsci.py:466-505 — the loss expressions transcribed term by term (this file is
        # out of the hot path's scope beyond GNNModel; kept so that fit() runs end to end)
        mask = self._mask(mask) if not (isinstance(mask, torch.Tensor) and mask.device == batch.device) else mask
        g = adj_orig.shape[0]
        deg = adj_orig.sum(axis=1)
        pos_weight = (g**2 - deg) / deg
        norm_adj = g * g / float((g * g - adj_orig.sum()) * 2)
        loss_adj = la * norm_adj * torch.mean(F.cross_entropy(z_adj, adj_orig, pos_weight))
        eps = 1e-10
        mean = mean * torch.reshape(self.size_factors, (-1, 1))
        disp = torch.clamp(disp, max=1e6)
        t1 = torch.lgamma(disp + eps) + torch.lgamma(batch + 1) - torch.lgamma(batch + disp + eps)
        t2 = (disp + batch) * torch.log(1.0 + (mean / (disp + eps))) + (batch * (torch.log(disp + eps) - torch.log(mean + eps)))
        nb_loss = t1 + t2
        nb_loss = torch.where(torch.isnan(nb_loss), torch.full_like(nb_loss, float("inf")), nb_loss)
        zero_nb = torch.pow(disp / (disp + mean + eps), disp)
        zero_case = -torch.log(pi + ((1 - pi) * zero_nb) + eps)
        loss_exp = torch.where(torch.lt(batch, 1e-8), zero_case, nb_loss)
        loss_exp = le * torch.mean(loss_exp[mask])
        log_lik = loss_exp + loss_adj
        kl_adj = (0.5 / batch.shape[0]) * torch.mean(
            torch.sum(1 + 2 * z_adj_log_std - torch.square(z_adj_mean) - torch.square(torch.exp(z_adj_log_std)), 1))
        kl_exp = 0.5 / batch.shape[1] * torch.mean(F.mse_loss(z_exp, batch, reduction="none")[mask])
        kl = ka * kl_adj - ke * kl_exp
        return loss_adj, loss_exp, log_lik, kl, log_lik - kl

    def load_model(self):
        state = torch.load(self.save_path / f"{self.dataset}.pt", map_location=self.device)
        self.aemodel.load_state_dict(state["aemodel"])
        self.gnnmodel.load_state_dict(state["gnnmodel"])

    def score(self, true_expr, imputed_expr, mask=None, metric="MSE", log1p=True, test_idx=None):
        """RMSE over the test cells, or PCC / MRE over their entries outside ``mask`` (graphsci.py:513-560; (sic) the default
        ``metric="MSE"`` is not an allowed option there either)."""
        allowd_metrics = {"RMSE", "PCC", "MRE"}
        if metric not in allowd_metrics:
            raise ValueError("scoring metric %r." % allowd_metrics)
        if test_idx is None:
            test_idx = range(len(true_expr))
        rows = torch.as_tensor(list(test_idx), device=self.device)
        true_target = true_expr.to(self.device)[rows]
        imputed_target = imputed_expr.to(self.device)[rows]
        if log1p:
            imputed_target = torch.log1p(imputed_target)
        if mask is not None:
            m = self._mask(mask)[rows]
            imputed_target[m] = true_target[m].to(imputed_target.dtype)
        if metric == "RMSE":
            return np.sqrt(F.mse_loss(true_target, imputed_target).item())
        actual, predicted = true_target[~m], imputed_target[~m]
        if metric == "PCC":
            return np.corrcoef(actual.cpu(), predicted.cpu())[0, 1]
        abs_actual = torch.abs(actual)
        abs_actual[abs_actual < 1e-10] = 1e-10
        return torch.mean(torch.abs(predicted - actual) / abs_actual).item()
