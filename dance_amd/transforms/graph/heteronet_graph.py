"""HeteronetGraph on MI355X — drop-in for dance/transforms/graph/heteronet_graph.py:14-94: directed kNN graph
i -> j for the k+1 nearest rows j of i (self included, :36-39), built with the exact HIP kNN."""
import numpy as np
import torch

from ... import kernels
from ...cellgraph import CellGeneGraph
from ...registry import register_preprocessor
from ..base import BaseTransform


@register_preprocessor("graph", "cell")
class HeteronetGraph(BaseTransform):

    def __init__(self, knn_num: int = 5, distance_metrics: str = "l2", random_state: int = 0, channel=None,
                 channel_type="X", ignore_first: bool = False, device="cuda", **kwargs):
        super().__init__(**kwargs)
        self.knn_num = knn_num
        self.distance_metrics = distance_metrics
        self.random_state = random_state
        self.channel = channel
        self.ignore_first = ignore_first
        self.channel_type = channel_type
        self.device = device

    def build_graph(self, features_np, radius=None, knears=None, distance_metrics="l2"):
        """Edge list [[i, j] ...] (int64 ndarray) in the reference's order: i ascending, j by (distance, index)."""
        if radius:
            raise NotImplementedError("radius graphs are not used on this path (knears only)")
        if distance_metrics not in ("l2", "euclidean", "minkowski", "cosine", "correlation"):
            raise NotImplementedError(f"distance metric {distance_metrics!r}: the HIP kNN ranks by l2, cosine or correlation distance")
        feats = np.ascontiguousarray(features_np, dtype=np.float32)
        if distance_metrics in ("cosine", "correlation"):
            # 1 - cos(u, v) = |u' - v'|^2 / 2 on the (centred and) l2-normalised rows: the same neighbours in the same order
            if distance_metrics == "correlation":
                feats = feats - feats.mean(1, keepdims=True)
            norm = np.linalg.norm(feats, axis=1, keepdims=True)
            norm[norm == 0] = 1
            feats = np.ascontiguousarray(feats / norm, dtype=np.float32)
        x = torch.from_numpy(feats).to(self.device)
        idx, _ = kernels.knn(x, knears + 1)
        idx = idx.cpu().numpy().astype(np.int64)
        n, k = idx.shape
        edges = np.stack((np.repeat(np.arange(n), k), idx.reshape(-1)), axis=1)
        return edges[edges[:, 1] >= 0]

    def __call__(self, data):
        adata = data.data
        features_np = data.get_feature(return_type="numpy", channel=self.channel, channel_type=self.channel_type)
        features = torch.as_tensor(features_np, dtype=torch.float32)
        num_nodes = features.shape[0]
        labels = torch.as_tensor(np.argmax(np.asarray(adata.obsm["cell_type"]), axis=1), dtype=torch.long)
        if self.ignore_first:
            labels[labels == 0] = -1
        edge_list = self.build_graph(features_np, knears=self.knn_num, distance_metrics=self.distance_metrics)
        src = torch.from_numpy(edge_list[:, 0]).to(self.device)
        dst = torch.from_numpy(edge_list[:, 1]).to(self.device)
        # CSR by destination with edge ids = positions in the reference edge list
        order = torch.argsort(dst * (edge_list.shape[0] + 1) + torch.arange(edge_list.shape[0], device=self.device))
        rowptr = torch.zeros(num_nodes + 1, dtype=torch.int64, device=self.device)
        rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=num_nodes), 0)
        g = CellGeneGraph(rowptr.to(torch.int32), src[order].to(torch.int32),
                          torch.ones(edge_list.shape[0], dtype=torch.float32, device=self.device), order.to(torch.int32),
                          num_nodes, {"feat": features.to(self.device), "label": labels.to(self.device)})
        batchs = adata.obs.get("batch_id", None) if hasattr(adata.obs, "get") else None
        if batchs is not None:
            g.ndata["batch_id"] = torch.from_numpy(np.asarray(batchs).astype(int)).long().to(self.device)
        adata.uns[self.out] = g
