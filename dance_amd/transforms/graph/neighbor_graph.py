"""NeighborGraph on MI355X — drop-in for dance/transforms/graph/neighbor_graph.py:9-57.

The reference wraps ``sc.pp.neighbors(...).obsp["connectivities"]``.  Here the same quantity is computed on the
GPU: exact brute-force kNN (dh_knn_bruteforce_f32, self counted as neighbour #0 like scanpy) followed by UMAP's
fuzzy-simplicial-set weights and the W + W^T - W o W^T symmetrisation (dh_umap_membership_f32 ...).  scanpy
switches to approximate NN-descent above ~4-8k cells; we stay exact at every size.
"""
import numpy as np
import scipy.sparse as sp
import torch

from ... import kernels
from ...graph import CSRGraph
from ...registry import register_preprocessor
from ..base import BaseTransform


@register_preprocessor("graph", "cell")
class NeighborGraph(BaseTransform):

    _DISPLAY_ATTRS = ("n_neighbors", "n_pcs", "knn", "random_state", "method", "metric")

    def __init__(self, n_neighbors: int = 15, *, n_pcs=None, knn: bool = True, random_state: int = 0,
                 method="umap", metric: str = "euclidean", channel="CellPCA", device="cuda", **kwargs):
        super().__init__(**kwargs)
        self.n_neighbors = n_neighbors
        self.n_pcs = n_pcs
        self.knn = knn
        self.random_state = random_state  # kept for repr/hash parity; the exact search is seed-free
        self.method = method
        self.metric = metric
        self.channel = channel
        self.device = device

    def _representation(self, data) -> np.ndarray:
        if self.channel is None:
            rep = data.get_feature(return_type="numpy", channel_type="X")
        else:
            rep = data.get_feature(return_type="numpy", channel=self.channel, channel_type="obsm")
        if self.n_pcs is not None:
            rep = rep[:, :self.n_pcs]
        rep = np.ascontiguousarray(rep, dtype=np.float32)
        if self.metric == "euclidean":
            return rep
        if self.metric in ("cosine", "correlation"):
            # both are monotone in the euclidean distance of (centred and) l2-normalised rows:
            # |u - v|^2 = 2 (1 - cos(u, v)); neighbour lists coincide, distances are rescaled below
            if self.metric == "correlation":
                rep = rep - rep.mean(1, keepdims=True)
            norm = np.linalg.norm(rep, axis=1, keepdims=True)
            norm[norm == 0] = 1
            return np.ascontiguousarray(rep / norm, dtype=np.float32)
        raise NotImplementedError(f"metric {self.metric!r} is not supported by the HIP kNN (euclidean/cosine/correlation)")

    def __call__(self, data):
        if self.method != "umap" or not self.knn:
            raise NotImplementedError("NeighborGraph on HIP implements method='umap', knn=True (the reference defaults)")
        self.logger.info("Start computing the kNN connectivity adjacency matrix")
        x = torch.from_numpy(self._representation(data)).to(self.device)
        idx, dist = kernels.knn(x, self.n_neighbors)
        if self.metric in ("cosine", "correlation"):
            dist = dist * dist * 0.5  # 1 - cos / 1 - corr
        (rowptr, col, val), _ = kernels.umap_connectivities(idx, dist.contiguous())
        n = x.shape[0]
        adj = sp.csr_matrix((val.cpu().numpy(), col.cpu().numpy(), rowptr.cpu().numpy()), shape=(n, n))
        data.data.obsp[self.out] = adj
        # device-resident copy for the GCN layers (value-symmetric: no transpose needed in backward)
        data.data.uns[f"{self.out}.hip"] = CSRGraph(rowptr, col, val, n, n, symmetric=True)
        return data
