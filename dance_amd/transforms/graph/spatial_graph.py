"""Spatial graph builders on MI355X — drop-ins for dance/transforms/graph/spatial_graph.py:13-151
(SpaGCNGraph, SpaGCNGraph2D, StagateGraph).  The O(N^2) numba ``pairwise_distance`` becomes the HIP kernel; the
per-spot histology window mean stays a (vectorisable) host loop exactly as in the reference — it is O(N) and
image-bound, not on the kernel path."""
import numpy as np
import scipy.sparse as sp
import torch

from ... import kernels
from ...registry import register_preprocessor
from ...utils.matrix import pairwise_distance
from ..base import BaseTransform


@register_preprocessor("graph", "spatial")
class SpaGCNGraph(BaseTransform):

    _DISPLAY_ATTRS = ("alpha", "beta")

    def __init__(self, alpha, beta, *, channels=("spatial", "spatial_pixel", "image"), channel_types=("obsm", "obsm", "uns"),
                 device="cuda", **kwargs):
        super().__init__(**kwargs)
        self.alpha = alpha
        self.beta = beta
        self.channels = channels
        self.channel_types = channel_types
        self.device = device

    def xyz(self, data) -> np.ndarray:
        """[x, y, z] per spot (spatial_graph.py:37-58): z = histology grey value, variance-weighted over colour
        channels, z-scored and scaled by max std(x, y) * alpha."""
        xy = data.get_feature(return_type="numpy", channel=self.channels[0], channel_type=self.channel_types[0])
        xy_pixel = data.get_feature(return_type="numpy", channel=self.channels[1], channel_type=self.channel_types[1])
        img = data.get_feature(return_type="numpy", channel=self.channels[2], channel_type=self.channel_types[2])
        g = np.zeros((xy.shape[0], 3))
        beta_half = round(self.beta / 2)
        x_lim, y_lim = img.shape[:2]
        for i, (x_pixel, y_pixel) in enumerate(xy_pixel):
            top, left = max(0, x_pixel - beta_half), max(0, y_pixel - beta_half)
            bottom, right = min(x_lim, x_pixel + beta_half + 1), min(y_lim, y_pixel + beta_half + 1)
            g[i] = np.mean(img[top:bottom, left:right], axis=(0, 1))
        g_var = g.var(0)
        z = (g * g_var).sum(1, keepdims=True) / g_var.sum()
        z = (z - z.mean()) / z.std()
        z *= xy.std(0).max() * self.alpha
        return np.hstack((xy, z)).astype(np.float32)

    def __call__(self, data):
        self.logger.info("Start calculating the adjacency matrix using the histology image")
        data.data.obsp[self.out] = pairwise_distance(self.xyz(data), dist_func_id=0, device=self.device)
        return data


@register_preprocessor("graph", "spatial")
class SpaGCNGraph2D(BaseTransform):

    def __init__(self, *, channel: str = "spatial_pixel", device="cuda", **kwargs):
        super().__init__(**kwargs)
        self.channel = channel
        self.device = device

    def __call__(self, data):
        x = data.get_feature(channel=self.channel, channel_type="obsm", return_type="numpy")
        data.data.obsp[self.out] = pairwise_distance(np.ascontiguousarray(x, dtype=np.float32), dist_func_id=0,
                                                     device=self.device)
        return data


@register_preprocessor("graph", "spatial")
class StagateGraph(BaseTransform):
    """STAGATE spatial graph: 0/1 CSR of the k nearest spots (``knn``) — exact HIP kNN — or of all spots within
    ``radius`` (sklearn radius search on the host, as in the reference; no HIP radius kernel yet)."""

    _MODELS = ("radius", "knn")
    _DISPLAY_ATTRS = ("model_name", "radius", "n_neighbors")

    def __init__(self, model_name: str = "radius", *, radius: float = 1, n_neighbors: int = 5, channel: str = "spatial_pixel",
                 channel_type: str = "obsm", device="cuda", **kwargs):
        super().__init__(**kwargs)
        if not isinstance(model_name, str) or (model_name.lower() not in self._MODELS):
            raise ValueError(f"Unknown model {model_name!r}, available options are {self._MODELS}")
        self.model_name = model_name
        self.radius = radius
        self.n_neighbors = n_neighbors
        self.channel = channel
        self.channel_type = channel_type
        self.device = device

    def __call__(self, data):
        xy_pixel = data.get_feature(return_type="numpy", channel=self.channel, channel_type=self.channel_type)
        n = xy_pixel.shape[0]
        if self.model_name.lower() == "radius":
            from sklearn.neighbors import NearestNeighbors
            adj = NearestNeighbors(radius=self.radius).fit(xy_pixel).radius_neighbors_graph(xy_pixel)
        else:
            x = torch.from_numpy(np.ascontiguousarray(xy_pixel, dtype=np.float32)).to(self.device)
            idx, _ = kernels.knn(x, self.n_neighbors)
            idx = idx.cpu().numpy()
            rows = np.repeat(np.arange(n), idx.shape[1])
            keep = idx.reshape(-1) >= 0
            adj = sp.csr_matrix((np.ones(int(keep.sum())), (rows[keep], idx.reshape(-1)[keep])), shape=(n, n))
            adj.sort_indices()
        data.data.obsp[self.out] = adj
