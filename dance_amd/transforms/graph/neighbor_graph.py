"""NeighborGraph on MI355X — drop-in for dance/transforms/graph/neighbor_graph.py:9-57.

The reference wraps ``sc.pp.neighbors(...).obsp["connectivities"]``.  Here the same quantity is computed on the
GPU: exact brute-force kNN (dh_knn_bruteforce_f32, self counted as neighbour #0 like scanpy) followed by UMAP's
fuzzy-simplicial-set weights and the W + W^T - W o W^T symmetrisation (dh_umap_membership_f32 ...).  scanpy
switches to approximate NN-descent above ~4-8k cells; we stay exact at every size.
"""
import numpy as np
import torch

from ... import kernels
from ...graph import CSRGraph, LazyScipyCSR
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

    def _representation(self, data) -> torch.Tensor:
        """The representation the neighbours are searched in, as an fp32 device tensor.  A slot that already lives on the device
        (DeviceArray, written by a device transform) is used in place — centring / normalising for the cosine-type metrics then
        run on the device too; a host slot goes through the numpy arithmetic it always did and is uploaded once."""
        from ...data import DeviceArray
        kw = dict(channel_type="X") if self.channel is None else dict(channel=self.channel, channel_type="obsm")
        if self.metric not in ("euclidean", "cosine", "correlation"):
            raise NotImplementedError(f"metric {self.metric!r} is not supported by the HIP kNN (euclidean/cosine/correlation)")
        self._device_in = isinstance(data.get_feature(return_type="default", **kw), DeviceArray)
        if self._device_in:
            rep = data.get_feature(return_type=self.device, **kw)
            if self.n_pcs is not None:
                rep = rep[:, :self.n_pcs]
            if self.metric != "euclidean":
                if self.metric == "correlation":
                    rep = rep - rep.mean(1, keepdim=True)
                norm = torch.linalg.vector_norm(rep, dim=1, keepdim=True)
                rep = rep / torch.where(norm == 0, torch.ones_like(norm), norm)
            return rep.contiguous()
        rep = data.get_feature(return_type="numpy", **kw)
        if self.n_pcs is not None:
            rep = rep[:, :self.n_pcs]
        rep = np.ascontiguousarray(rep, dtype=np.float32)
        if self.metric != "euclidean":
            # cosine / correlation are monotone in the euclidean distance of (centred and) l2-normalised rows:
            # |u - v|^2 = 2 (1 - cos(u, v)); neighbour lists coincide, distances are rescaled in __call__
            if self.metric == "correlation":
                rep = rep - rep.mean(1, keepdims=True)
            norm = np.linalg.norm(rep, axis=1, keepdims=True)
            norm[norm == 0] = 1
            rep = np.ascontiguousarray(rep / norm, dtype=np.float32)
        return torch.from_numpy(rep).to(self.device)

    def __call__(self, data):
        if self.method != "umap" or not self.knn:
            raise NotImplementedError("NeighborGraph on HIP implements method='umap', knn=True (the reference defaults)")
        self.logger.info("Start computing the kNN connectivity adjacency matrix")
        x = self._representation(data)
        idx, dist = kernels.knn(x, self.n_neighbors)
        if self.metric in ("cosine", "correlation"):
            dist = dist * dist * 0.5  # 1 - cos / 1 - corr
        (rowptr, col, val), _ = kernels.umap_connectivities(idx, dist.contiguous())
        n = x.shape[0]
        # the device-resident graph is what the GCN layers consume (value-symmetric: no transpose needed in backward).  The obsp slot
        # holds the scipy matrix the reference stores there; inside an on-device pipeline (the representation came from a
        # DeviceArray) it is built on first access only (LazyScipyCSR), so the pipeline itself copies nothing to the host
        g = CSRGraph(rowptr, col, val, n, n, symmetric=True)
        data.data.uns[f"{self.out}.hip"] = g
        data.data.obsp[self.out] = LazyScipyCSR(g) if self._device_in else g.to_scipy()
        return data
