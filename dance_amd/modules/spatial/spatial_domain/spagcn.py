"""SpaGCN's graph convolution on MI355X — drop-in for ``GraphConvolution`` of
dance/modules/spatial/spatial_domain/spagcn.py:337-366 (parameters ``weight``/``bias``, U(+-1/sqrt(out)) init,
``forward(input, adj)``).

The reference multiplies by a DENSE N x N adjacency (spagcn.py:497,359).  A dense ``adj`` tensor is honoured
with the MFMA GEMM (exact reference arithmetic, small N); a sparse tensor / ``CSRGraph`` (the kNN-truncated
Gaussian kernel used at scale) goes through the CSR SpMM.
"""
import logging

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn, optim
from torch.nn.parameter import Parameter

from .... import kernels
from ....autograd import dec_kl_loss, dec_target_distribution, dense_adj_layer, gcn_layer, student_t_assign
from ....graph import CSRGraph, as_graph
from ....sharding import ShardedGCNGraph, sharded_gcn_layer
from ....transforms import CellPCA, Compose, SetConfig
from ....transforms.graph import SpaGCNGraph, SpaGCNGraph2D
from ...base import BaseClusteringMethod

logger = logging.getLogger("dance")


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
        if isinstance(adj, ShardedGCNGraph):  # destination-range shard of the graph (BASELINE config 5): input = this rank's rows
            return sharded_gcn_layer(input, self.weight, adj, self.bias, False, ops=getattr(adj, "ops", None))
        if isinstance(adj, torch.Tensor) and adj.layout == torch.strided:
            return dense_adj_layer(input, self.weight, adj, self.bias)
        graph = adj if isinstance(adj, CSRGraph) else as_graph(adj, input.device)
        return gcn_layer(input, self.weight, graph, self.bias, False)

    def __repr__(self):
        return f"{self.__class__.__name__}({self.in_features} -> {self.out_features})"


def _on_gpu(device) -> bool:
    return str(device).startswith("cuda")


def _step(optimizer):
    """``optimizer.step()``; a ``torch.optim.Adam`` with device-side step counters (``capturable=True``: what the fits create on a GPU) takes
    dh_adam_step_f32 — two launches for gc.weight, gc.bias and mu instead of the ~14 of the framework's multi-tensor implementation — once
    its state exists (the first step creates it); torch's own arithmetic (adam.hip)."""
    if not kernels.adam_step(optimizer):
        optimizer.step()


def _to_device_f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def _embed_rows_aligned(x: torch.Tensor) -> torch.Tensor:
    """The embedding with every row starting on a 256-byte boundary (a [N, d] view of a zero-padded [N, 64] buffer, d <= 64): the
    fused narrow layer gathers whole rows, and a 50-float row at stride 50 straddles 2-3 cache lines (333 bytes fetched for 200 used,
    measured: 4.8 GB of traffic for 3.1 GB gathered) where an aligned one is exactly two."""
    if x.dim() != 2 or x.shape[1] > 64 or x.shape[1] == 64 or not x.is_cuda:
        return x
    buf = torch.zeros((x.shape[0], 64), dtype=torch.float32, device=x.device)
    buf[:, :x.shape[1]] = x
    return buf[:, :x.shape[1]]


def calculate_p(adj, l, device="cuda"):
    """spagcn.py:249-251: mean_i sum_j exp(-adj_ij^2 / (2 l^2)) - 1.  One streaming pass over the distance matrix
    (dh_gaussian_kernel_f32 row sums + dh_colsum_f32); the N x N kernel matrix is never materialised."""
    d = _to_device_f32(adj, device)
    _, rs = kernels.gaussian_kernel(d, l, want_out=False, want_rowsum=True)
    return float(kernels.colsum(rs[:, None])[0]) / d.shape[0] - 1


def search_l(p, adj, start=0.01, end=1000, tol=0.01, max_run=100, device="cuda"):
    """spagcn.py:254-287 bisection on l; the distance matrix is moved to the device once."""
    d = _to_device_f32(adj, device)
    run = 0
    p_low, p_high = calculate_p(d, start, device), calculate_p(d, end, device)
    if p_low > p + tol:
        logger.info("l not found, try smaller start point.")
        return None
    elif p_high < p - tol:
        logger.info("l not found, try bigger end point.")
        return None
    elif np.abs(p_low - p) <= tol:
        return start
    elif np.abs(p_high - p) <= tol:
        return end
    while (p_low + tol) < p < (p_high - tol):
        run += 1
        if run > max_run:
            logger.info(f"Exact l not found, closest values are:\nl={start}: p={p_low}\nl={end}: p={p_high}")
            return None
        mid = (start + end) / 2
        p_mid = calculate_p(d, mid, device)
        if np.abs(p_mid - p) <= tol:
            logger.info(f"recommended l: {mid}")
            return mid
        if p_mid <= p:
            start, p_low = mid, p_mid
        else:
            end, p_high = mid, p_mid
    return None


class SimpleGCDEC(nn.Module):
    """Basic model used in SpaGCN training (spagcn.py:369-584): GraphConvolution + DEC clustering head."""

    # measurement aid (scripts/bench_configs.py): True -> ``epoch_ms`` holds the device time of every iteration of the last fit_with_init
    record_epoch_times = False

    def __init__(self, nfeat, nhid, alpha=0.2, device="cuda"):
        super().__init__()
        self.gc = GraphConvolution(nfeat, nhid)
        self.nhid = nhid
        self.alpha = alpha
        self.device = device

    def forward(self, x, adj):
        x = self.gc(x, adj)
        # Student-t kernel exactly as written in the reference (:394-396), including the q**(alpha+1)/2 precedence — on the fused
        # kernel pair (no [N, C, d] broadcast tensor: it was 4.1 of an iteration's 4.8 ms at 500k spots)
        q = student_t_assign(x, self.mu, a=self.alpha, eps=1e-8, pw=self.alpha + 1.0, scale=0.5)
        return x, q

    # Sharded training (``adj`` is a ``ShardedGCNGraph``; one process per GPU): X, q, p are this rank's rows; the reductions
    # over all spots below are completed with all-reduces, so the arithmetic is that of the single-process model.
    _sg = None

    def _allsum(self, t):
        if self._sg is not None and self._sg.world > 1:
            import torch.distributed as dist
            t = t.clone()
            dist.all_reduce(t, group=self._sg.group)
        return t

    def loss_function(self, p, q):
        # kld(p, q) = mean over spots of sum_j p log(p / (q + 1e-6)) (:399-407); sharded: the local sum over N_all (its gradient is the local
        # part of the global mean's; mu's is all-reduced in fit).  One fused kernel each way instead of ~10 elementwise launches.
        n_all = q.shape[0] if self._sg is None else self._sg.n_nodes
        return dec_kl_loss(p, q, eps=1e-6, scale=1.0 / n_all)

    def target_distribution(self, q):
        if self._sg is not None and self._sg.world > 1:
            return dec_target_distribution(q.detach(), self._allsum(torch.sum(q.detach(), dim=0)))
        return dec_target_distribution(q.detach()) if not q.requires_grad or not torch.is_grad_enabled() else dec_target_distribution(q)

    def _adj(self, adj):
        if isinstance(adj, (CSRGraph, ShardedGCNGraph)):
            return adj
        if isinstance(adj, torch.Tensor) and adj.layout != torch.strided:
            return as_graph(adj, self.device)
        return _to_device_f32(adj, self.device)

    def fit(self, X, adj, lr=0.001, epochs=5000, update_interval=3, trajectory_interval=50, weight_decay=5e-4, opt="sgd",
            init="louvain", n_neighbors=10, res=0.4, n_clusters=10, init_spa=True, tol=1e-3):
        self.trajectory = []
        self.to(self.device)
        X = _embed_rows_aligned(_to_device_f32(X, self.device))
        adj = self._adj(adj)
        self._sg = adj if isinstance(adj, ShardedGCNGraph) else None
        sharded = self._sg is not None and self._sg.world > 1
        if self._sg is not None:
            import torch.distributed as dist
            lo, hi = self._sg.ranges[self._sg.rank]
            if X.shape[0] == self._sg.n_nodes:
                X = X[lo:hi].contiguous()  # the caller handed the whole matrix: keep this rank's rows
            if sharded:
                for t in self.gc.parameters():
                    dist.broadcast(t.data, src=0, group=self._sg.group)
        if opt == "sgd":
            optimizer = optim.SGD(self.parameters(), lr=lr, momentum=0.9)
        elif opt == "admin":
            optimizer = optim.Adam(self.parameters(), lr=lr, weight_decay=weight_decay, capturable=_on_gpu(self.device))
        else:
            raise ValueError(f"Unknown optimizer {opt!r}")
        with torch.no_grad():
            features = self.gc(X, adj)
        feats_all, x_all = features, X
        if sharded:  # the initial clustering sees all spots: gather (N x nhid floats), cluster on rank 0, broadcast the labels
            feats_all = self._sg.all_gather_rows(features)[:self._sg.n_nodes]
            x_all = self._sg.all_gather_rows(X)[:self._sg.n_nodes]
        if init == "kmeans":
            from sklearn.cluster import KMeans
            self.n_clusters = n_clusters
            kmeans = KMeans(self.n_clusters, n_init=20)
            y_pred = kmeans.fit_predict(feats_all.cpu().numpy() if init_spa else x_all.cpu().numpy())
        elif init == "louvain":
            # spagcn.py:480-492: sc.pp.neighbors(n_neighbors) + sc.tl.leiden(resolution=res).  Neighbour graph on the
            # GPU (exact kNN + UMAP connectivities); the Leiden algorithm itself runs on the host (dance_amd/utils/community.py:
            # scanpy / leidenalg are not installable — same algorithm and quality function, not leidenalg's random stream)
            from ....utils.community import leiden_like
            logger.info(f"Initializing cluster centers with louvain, resolution = {res}")
            y_pred = leiden_like(feats_all if init_spa else x_all, n_neighbors, resolution=res, device=self.device)
            self.n_clusters = len(np.unique(y_pred))
        else:
            raise ValueError(f"Unknown init {init!r}")
        if sharded:
            yb = torch.from_numpy(np.asarray(y_pred, dtype=np.int64)).to(self.device)
            dist.broadcast(yb, src=0, group=self._sg.group)
            y_pred = yb.cpu().numpy()
            self.n_clusters = int(yb.max()) + 1
        yt_all = torch.from_numpy(np.asarray(y_pred)).to(self.device)
        centers = torch.stack([feats_all[yt_all == c].mean(0) for c in range(self.n_clusters)])  # groupby("Group").mean()
        if self._sg is not None:
            y_pred = y_pred[lo:hi]  # from here on: this rank's spots
        y_pred_last = y_pred
        self.mu = Parameter(torch.empty(self.n_clusters, self.nhid, device=self.device))
        self.trajectory.append(y_pred)
        self.mu.data.copy_(centers)
        if opt == "sgd":  # mu was created after the optimizer in the reference too (:494), hence not optimised
            pass
        self.train()
        for epoch in range(epochs):
            fwd = None
            if epoch % update_interval == 0:
                # the reference runs the forward twice here — once for the target (:512-514), once for the loss (:516) — with nothing
                # changed in between and no randomness in SimpleGCDEC's forward: the second pass is the first, bit for bit, so ONE is run
                fwd = self(X, adj)
                p = self.target_distribution(fwd[1].detach()).data
            optimizer.zero_grad()
            z, q = fwd if fwd is not None else self(X, adj)
            loss = self.loss_function(p, q)
            loss.backward()
            if sharded and self.mu.grad is not None:  # gc's gradients are all-reduced inside the sharded layer; mu's here
                dist.all_reduce(self.mu.grad, group=self._sg.group)
            _step(optimizer)
            if epoch % trajectory_interval == 0:
                self.trajectory.append(torch.argmax(q, dim=1).data.cpu().numpy())
            y_pred = torch.argmax(q, dim=1).data.detach().cpu().numpy()
            changed = torch.tensor([float(np.sum(y_pred != y_pred_last))], device=self.device)
            delta_label = np.float32(float(self._allsum(changed)) / (self._sg.n_nodes if self._sg is not None else X.shape[0]))
            y_pred_last = y_pred
            if epoch > 0 and (epoch - 1) % update_interval == 0 and delta_label < tol:
                logger.info(f"delta_label {delta_label} < tol {tol}; total epoch: {epoch}")
                break

    def fit_with_init(self, X, adj, init_y, lr=0.001, epochs=5000, update_interval=1, weight_decay=5e-4, opt="sgd"):
        """spagcn.py:541-584: cluster centres = per-group means of the initial embedding under the GIVEN labels ``init_y``
        (``groupby("Group").mean()``: groups in sorted label order), then the DEC loop without early stopping.  The
        optimiser is created before ``mu`` receives its values but — unlike ``fit`` — over an existing ``mu`` parameter when
        the model has one, exactly as ``self.parameters()`` yields it there."""
        self.to(self.device)
        X = _embed_rows_aligned(_to_device_f32(X, self.device))  # as in fit: whole 256-byte rows for the fused narrow layer's gathers
        adj = self._adj(adj)
        if opt == "sgd":
            optimizer = optim.SGD(self.parameters(), lr=lr, momentum=0.9)
        elif opt == "admin":
            optimizer = optim.Adam(self.parameters(), lr=lr, weight_decay=weight_decay, capturable=_on_gpu(self.device))
        else:
            raise ValueError(f"Unknown optimizer {opt!r}")
        with torch.no_grad():
            features = self.gc(X, adj)
        labels = np.asarray(init_y)
        groups = np.unique(labels)  # pandas groupby sorts its keys
        yt = torch.from_numpy(np.searchsorted(groups, labels)).to(self.device)
        centers = torch.stack([features[yt == c].mean(0) for c in range(len(groups))])
        if getattr(self, "mu", None) is None:  # the reference copies into an existing self.mu (set by an earlier fit)
            self.n_clusters = len(groups)
            self.mu = Parameter(torch.empty(self.n_clusters, self.nhid, device=self.device))
        self.mu.data.copy_(centers)
        self.train()
        marks = []
        for epoch in range(epochs):
            if self.record_epoch_times and torch.cuda.is_available():  # measurement aid: see the class attribute
                marks.append(torch.cuda.Event(enable_timing=True))
                marks[-1].record()
            fwd = None
            if epoch % update_interval == 0:
                # the reference runs the forward twice here — once for the target (:512-514), once for the loss (:516) — with nothing
                # changed in between and no randomness in SimpleGCDEC's forward: the second pass is the first, bit for bit, so ONE is run
                fwd = self(X, adj)
                p = self.target_distribution(fwd[1].detach()).data
            optimizer.zero_grad()
            z, q = fwd if fwd is not None else self(X, adj)
            loss = self.loss_function(p, q)
            loss.backward()
            _step(optimizer)
        if marks:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
            marks[-1].synchronize()
            self.epoch_ms = [a.elapsed_time(b) for a, b in zip(marks[:-1], marks[1:])]

    @torch.no_grad()
    def predict(self, X, adj):
        X, adj = _to_device_f32(X, self.device), self._adj(adj)
        if isinstance(adj, ShardedGCNGraph):  # this rank's rows in, ALL rows out (gathered), like the single-process call
            lo, hi = adj.ranges[adj.rank]
            z, q = self(X[lo:hi].contiguous() if X.shape[0] == adj.n_nodes else X, adj)
            return adj.all_gather_rows(z)[:adj.n_nodes], adj.all_gather_rows(q)[:adj.n_nodes]
        return self(X, adj)


class GC_DEC(SimpleGCDEC):
    """spagcn.py:588-697: the DEC model with TWO graph convolutions (ReLU + dropout in between) — defined next to SimpleGCDEC in the
    reference, not instantiated by ``SpaGCN``.  Each convolution is the fused GCN layer (ReLU in the SpMM epilogue, bitmap-masked
    backward); the Student-t head, the losses and the two training loops are SimpleGCDEC's (the reference's GC_DEC restates them with
    ``1e-6`` instead of ``1e-8`` in the kernel's denominator, :605).  ``mu`` exists from construction, as in the reference."""

    def __init__(self, nfeat, nhid1, nhid2, n_clusters=None, dropout=0.5, alpha=0.2, device="cuda"):
        nn.Module.__init__(self)
        self.gc1 = GraphConvolution(nfeat, nhid1)
        self.gc2 = GraphConvolution(nhid1, nhid2)
        self.dropout = dropout
        self.mu = Parameter(torch.empty(n_clusters, nhid2))
        self.n_clusters, self.nhid, self.alpha, self.device = n_clusters, nhid2, alpha, device

    def gc(self, x, adj):
        """The embedding both training loops start from (SimpleGCDEC calls its single layer ``gc``)."""
        x = F.relu(self.gc1(x, adj))
        x = F.dropout(x, self.dropout, training=True)  # training=True always, as in the reference (:602)
        return self.gc2(x, adj)

    def forward(self, x, adj):
        x = self.gc(x, adj)
        q = student_t_assign(x, self.mu, a=self.alpha, eps=1e-6, pw=self.alpha + 1.0, scale=0.5)   # :605-607
        return x, q

    def fit(self, X, adj, lr=0.001, epochs=10, update_interval=5, weight_decay=5e-4, opt="sgd", init="louvain", n_neighbors=10, res=0.4):
        """:622-670: initial clustering of the embedding (k-means with ``n_clusters`` centres, or the community detection), centres =
        group means, then the DEC loop without early stopping, the prediction of every epoch kept in ``trajectory``."""
        self.to(self.device)
        X, adj = _embed_rows_aligned(_to_device_f32(X, self.device)), self._adj(adj)
        self.trajectory = []
        optimizer = self._optimizer(opt, lr, weight_decay)
        with torch.no_grad():
            features = self.gc(X, adj)
        if init == "kmeans":
            from sklearn.cluster import KMeans
            y_pred = KMeans(self.n_clusters, n_init=20).fit_predict(features.cpu().numpy())
        elif init == "louvain":
            from ....utils.community import leiden_like
            y_pred = np.asarray(leiden_like(features, n_neighbors, resolution=res, device=self.device))
        else:
            raise ValueError(f"Unknown init {init!r}")
        self.trajectory.append(y_pred)
        self._centres_from(features, y_pred)
        self.train()
        for epoch in range(epochs):
            if epoch % update_interval == 0:
                with torch.no_grad():  # (the reference builds and drops an autograd graph here; the values are the same)
                    _, q = self.forward(X, adj)
                p = self.target_distribution(q).data
            optimizer.zero_grad()
            z, q = self(X, adj)
            loss = self.loss_function(p, q)
            loss.backward()
            _step(optimizer)
            self.trajectory.append(torch.argmax(q, dim=1).data.cpu().numpy())

    def _optimizer(self, opt, lr, weight_decay):
        if opt == "sgd":
            return optim.SGD(self.parameters(), lr=lr, momentum=0.9)
        if opt == "admin":
            return optim.Adam(self.parameters(), lr=lr, weight_decay=weight_decay, capturable=_on_gpu(self.device))
        raise ValueError(f"Unknown optimizer {opt!r}")

    def _centres_from(self, features, labels):
        labels = np.asarray(labels)
        groups = np.unique(labels)  # pandas groupby sorts its keys
        yt = torch.from_numpy(np.searchsorted(groups, labels)).to(features.device)
        self.mu.data.copy_(torch.stack([features[yt == c].mean(0) for c in range(len(groups))]))

    def fit_with_init(self, X, adj, init_y, lr=0.001, epochs=10, update_interval=1, weight_decay=5e-4, opt="sgd"):
        """:672-697."""
        self.to(self.device)
        X, adj = _embed_rows_aligned(_to_device_f32(X, self.device)), self._adj(adj)
        optimizer = self._optimizer(opt, lr, weight_decay)
        with torch.no_grad():
            features = self.gc(X, adj)
        self._centres_from(features, init_y)
        self.train()
        for epoch in range(epochs):
            if epoch % update_interval == 0:
                with torch.no_grad():  # (the reference builds and drops an autograd graph here; the values are the same)
                    _, q = self.forward(X, adj)
                p = self.target_distribution(q).data
            optimizer.zero_grad()
            z, q = self(X, adj)
            loss = self.loss_function(p, q)
            loss.backward()
            _step(optimizer)


def refine(sample_id, pred, dis, shape="hexagon"):
    """spagcn.py:290-334: majority vote among the nearest spots (6 for hexagon, 4 for square) when more than half of
    them disagree with the spot's own label.  The neighbour lists come from a device top-k of the distance rows
    instead of a pandas sort per spot."""
    num_nbs = {"hexagon": 6, "square": 4}.get(shape)
    if num_nbs is None:
        raise ValueError("Shape not recongized, shape='hexagon' for Visium data, 'square' for ST data.")
    pred = np.asarray(pred)
    d = torch.as_tensor(np.asarray(dis), dtype=torch.float32)
    order = torch.argsort(d, dim=1, stable=True)[:, :num_nbs + 1].numpy()  # includes the spot itself
    refined = []
    for i in range(len(sample_id)):
        nbs_pred = pred[order[i]]
        self_pred = pred[i]
        vals, counts = np.unique(nbs_pred, return_counts=True)
        if counts[vals == self_pred][0] < num_nbs / 2 and counts.max() > num_nbs / 2:
            refined.append(vals[counts.argmax()])
        else:
            refined.append(self_pred)
    return refined


class SpaGCN(BaseClusteringMethod):
    """SpaGCN (spagcn.py:700-892): ``x = (embed, adj)`` with ``adj`` the distance matrix from SpaGCNGraph (dense
    ndarray / device tensor) or a kNN-truncated ``CSRGraph`` of distances for large N."""

    def __init__(self, l=None, device="cuda"):
        self.l = l
        self.res = None
        self.device = device

    @staticmethod
    def preprocessing_pipeline(alpha: float = 1, beta: int = 49, dim: int = 50, log_level="INFO"):
        """spagcn.py:715-731: drop spike-in / mitochondrial genes by name, normalize_total to 1e4 and log1p on the device, the two
        spatial graphs, and the PCA of the normalised matrix (on the device)."""
        from ....transforms import FilterGenesMatch, Log1P, NormalizeTotal
        return Compose(
            FilterGenesMatch(prefixes=["ERCC", "MT-"]),
            NormalizeTotal(target_sum=1e4, max_fraction=1.0),
            Log1P(),
            SpaGCNGraph(alpha=alpha, beta=beta),
            SpaGCNGraph2D(),
            CellPCA(n_components=dim, device="cuda"),
            SetConfig({
                "feature_channel": ["CellPCA", "SpaGCNGraph", "SpaGCNGraph2D"],
                "feature_channel_type": ["obsm", "obsp", "obsp"],
                "label_channel": "label",
                "label_channel_type": "obs"
            }),
            log_level=log_level,
        )

    def search_l(self, p, adj, start=0.01, end=1000, tol=0.01, max_run=100):
        return search_l(p, adj, start, end, tol, max_run, device=self.device)

    def set_l(self, l):
        self.l = l

    def search_set_res(self, x, l, target_num, start=0.4, step=0.1, tol=5e-3, lr=0.05, epochs=10, max_run=10):
        """spagcn.py:771-805: search the clustering resolution that yields ``target_num`` clusters — short fits of fresh models
        at res +- step, halving the step when the direction flips.  (``self.res`` is only set on the path that falls out of the
        loop, as in the reference.)"""
        # reference: dance/modules/spatial/spatial_domain/spagcn.py:771-805 — host control flow transcribed (same bisection, stopping rule and log
        # strings) so that the recommended resolution is the reference's; the fits it launches are this package's
        res = start
        logger.info(f"Start at {res = :.4f}, {step = :.4f}")
        fit_kw = dict(init_spa=True, init="louvain", tol=tol, lr=lr, epochs=epochs)
        old_num = len(set(SpaGCN(l, device=self.device).fit_predict(x, res=res, **fit_kw)))
        logger.info(f"Res = {res:.4f}, Num of clusters = {old_num}")
        run = 0
        while old_num != target_num:
            old_sign = 1 if (old_num < target_num) else -1
            new_num = len(set(SpaGCN(l, device=self.device).fit_predict(x, res=res + step * old_sign, **fit_kw)))
            logger.info(f"Res = {res + step * old_sign:.3e}, Num of clusters = {new_num}")
            if new_num == target_num:
                res = res + step * old_sign
                logger.info(f"recommended res = {res:.4f}")
                return res
            new_sign = 1 if (new_num < target_num) else -1
            if new_sign == old_sign:
                res = res + step * old_sign
                logger.info(f"Res changed to {res}")
                old_num = new_num
            else:
                step = step / 2
                logger.info(f"Step changed to {step:.4f}")
            if run > max_run:
                logger.info(f"Exact resolution not found. Recommended res = {res:.4f}")
                return res
            run += 1
        logger.info(f"Recommended res = {res:.4f}")
        self.res = res
        return res

    def calc_adj_exp(self, adj):
        """exp(-adj^2 / (2 l^2)) on the device; a ``CSRGraph`` of distances keeps its sparsity pattern."""
        if isinstance(adj, ShardedGCNGraph):  # shards of the distance graph -> shards of the kernel graph (same structure, same halo plan)
            import copy

            from ....sharding import GraphShard
            out = copy.copy(adj)
            kern = lambda sh: GraphShard(sh.rowptr, sh.col, kernels.gaussian_kernel(sh.val, self.l)[0], sh.lo, sh.hi, sh.n_cols)
            out.a, out.at = kern(adj.a), kern(adj.at)
            if adj.full is not None:
                out.full = (kern(adj.full[0]), kern(adj.full[1]))
            out._local_graph = None
            return out
        if isinstance(adj, CSRGraph):
            vals, _ = kernels.gaussian_kernel(adj.val, self.l)  # flat launch over the nnz values
            return CSRGraph(adj.rowptr, adj.col, vals, adj.n_rows, adj.n_cols, symmetric=adj.symmetric)
        out, _ = kernels.gaussian_kernel(_to_device_f32(adj, self.device), self.l)
        return out

    def fit(self, x, y=None, *, num_pcs=50, lr=0.005, epochs=2000, weight_decay=0, opt="admin", init_spa=True,
            init="louvain", n_neighbors=10, n_clusters=None, res=0.4, tol=1e-3):
        embed, adj = x
        self.num_pcs, self.res, self.lr, self.epochs = num_pcs, res, lr, epochs
        self.weight_decay, self.opt, self.init_spa, self.init = weight_decay, opt, init_spa, init
        self.n_neighbors, self.n_clusters, self.tol = n_neighbors, n_clusters, tol
        if self.l is None:
            raise ValueError("l should be set before fitting the model!")
        self.model = SimpleGCDEC(embed.shape[1], embed.shape[1], device=self.device)
        adj_exp = self.calc_adj_exp(adj)
        self.model.fit(embed, adj_exp, lr=self.lr, epochs=self.epochs, weight_decay=self.weight_decay, opt=self.opt,
                       init_spa=self.init_spa, init=self.init, n_neighbors=self.n_neighbors, n_clusters=self.n_clusters,
                       res=self.res, tol=self.tol)

    def predict_proba(self, x):
        embed, adj = x
        _, pred_prob = self.model.predict(embed, self.calc_adj_exp(adj))
        return pred_prob

    def predict(self, x):
        return torch.argmax(self.predict_proba(x), dim=1).data.cpu().numpy()
