"""FeatureFeatureGraph on MI355X — drop-in for dance/transforms/graph/feature_feature_graph.py:14-87: the gene–gene similarity
graph GraphSCI convolves over (graphsci.py:170-199).  The similarity matrix is a Gram product over the cells, G x G with K = N:

* ``pearson``  — columns standardised on the device (``col_moments`` -> ``col_standardize`` with std * sqrt(N)), then ONE
  transposed exact-fp32 GEMM  Z^T Z  (row-reducing product: the split-K path of ``gemm_f32.hip``); clipped to [-1, 1] as
  ``np.corrcoef`` does;
* ``spearman`` — the same on the per-gene average ranks (ties share their mean rank, as ``scipy.stats.rankdata``);
* ``rbf``      — squared norms + the same Gram product -> distances -> ``exp(-d / denom)`` (``dist_to_rbf``, utils/matrix.py:70-97).

Entries with |s| < threshold are zeroed, the non-zeros listed in row-major order by ``dense_to_csr`` (the order
``scipy.sparse.coo_matrix`` lists them in, which is the reference's edge order), and — ``normalize_edges`` — each edge weighted
1 / sqrt(out_degree(src) * in_degree(dst)) (``dgl.nn.EdgeWeightNorm`` on unit weights) from the degree-scale kernel.

The result in ``data.data.uns[out]`` is a ``CSRGraph`` (rows = destination genes; the matrix is symmetric) with
``ndata["feat"]`` = the G x N gene features and ``edata["weight"]`` (= ``val``), which is what
``modules.single_modality.imputation.graphsci.GNNModel.forward`` takes.  Numerics: the reference correlates in float64 and casts
to float32 before thresholding; here the Gram product is exact-fp32 on the matrix cores, so a similarity within ~1e-6 of the
threshold may fall on the other side."""
from typing import Any, Dict, Optional

import numpy as np
import torch

from ... import kernels
from ...graph import CSRGraph
from ...registry import register_preprocessor
from ..base import BaseTransform


def average_ranks(x: torch.Tensor) -> torch.Tensor:
    """Per-column ranks 1..N with ties sharing their mean rank (``scipy.stats.rankdata(method="average")`` down every column),
    float64, by one sort per column and two scans over the tie-run boundaries."""
    n = x.shape[0]
    v, order = torch.sort(x, dim=0, stable=True)
    pos = torch.arange(n, device=x.device, dtype=torch.int64)[:, None].expand_as(v)
    new = torch.ones_like(v, dtype=torch.bool)
    new[1:] = v[1:] != v[:-1]
    first = torch.cummax(torch.where(new, pos, torch.zeros_like(pos)), dim=0).values          # start of each element's tie run
    ends = torch.ones_like(new)
    ends[:-1] = new[1:]
    last = torch.flip(torch.cummin(torch.flip(torch.where(ends, pos, torch.full_like(pos, n)), [0]), dim=0).values, [0])
    rank_sorted = (first + last).double() / 2 + 1
    return torch.empty_like(rank_sorted).scatter_(0, order, rank_sorted)


def correlation_matrix(x: torch.Tensor) -> torch.Tensor:
    """G x G Pearson correlation of the columns of the N x G device matrix ``x`` (float32 out)."""
    n = x.shape[0]
    s, q = kernels.col_moments(x)
    mean = s.double() / n
    var = (q.double() / n - mean * mean).clamp(min=0)
    z = kernels.col_standardize(x, mean, (var * n).sqrt())  # float64 statistics, fp32 result: Z^T Z is the correlation itself
    return kernels.gemm(z, z, trans_a=True, mode="exact").clamp_(-1.0, 1.0)


@register_preprocessor("graph", "feature")
class FeatureFeatureGraph(BaseTransform):

    _DISPLAY_ATTRS = ("threshold", "positive_only", "normalize_edges", "score_func", "score_func_kwargs")

    def __init__(self, threshold: float = 0.3, *, positive_only: bool = False, normalize_edges: bool = True, score_func="pearson",
                 score_func_kwargs: Optional[Dict[str, Any]] = None, device="cuda", **kwargs):
        super().__init__(**kwargs)
        self.threshold = threshold
        self.positive_only = positive_only
        self.normalize_edges = normalize_edges
        self.score_func = score_func
        self.score_func_kwargs = score_func_kwargs or {}
        self.device = device

    def _rbf(self, x: torch.Tensor, denom_scale: float = 1.0, scale_mode: str = "med_dist") -> torch.Tensor:
        sq = kernels.col_moments(x)[1].float()
        dist = (sq[None, :] + sq[:, None] - 2 * kernels.gemm(x, x, trans_a=True, mode="exact")).clamp_(min=0).sqrt_()
        if scale_mode == "med_dist":   # numpy's median: the mean of the two middle values of the G^2 entries
            flat = dist.reshape(-1).double()
            k = flat.numel()
            denom = (torch.kthvalue(flat, (k - 1) // 2 + 1).values + torch.kthvalue(flat, k // 2 + 1).values) / 2 * denom_scale
        elif scale_mode == "ind_med_dist":
            g = dist.shape[1]
            d = dist.double()
            denom = ((torch.kthvalue(d, (g - 1) // 2 + 1, dim=1).values + torch.kthvalue(d, g // 2 + 1, dim=1).values) / 2 * denom_scale)[:, None]
        elif scale_mode == "scale":
            denom = denom_scale
        else:
            raise ValueError(f"Uknwon rbf scaling mode {scale_mode}")
        return torch.exp(-dist.double() / denom).float()

    def __call__(self, data):
        feat = data.get_feature(return_type=self.device)  # the configured feature channel, like the reference's bare get_feature()
        if self.score_func == "pearson":
            adj = correlation_matrix(feat)
        elif self.score_func == "spearman":
            adj = correlation_matrix(average_ranks(feat).float())
        elif self.score_func == "rbf":
            adj = self._rbf(feat, **self.score_func_kwargs)
        else:
            raise ValueError(f"Unknown similarity score function {self.score_func!r}, "
                             "supported options are: 'pearson', 'spearman', 'rbf'")
        adj[(adj > -self.threshold) & (adj < self.threshold)] = 0
        if self.positive_only:
            adj[adj < 0] = 0
        n = adj.shape[0]
        rowptr, col, _ = kernels.dense_to_csr(adj)
        weight = torch.ones(col.numel(), dtype=torch.float32, device=adj.device)
        if self.normalize_edges and col.numel():
            rowscale, colscale = kernels.degree_scales(rowptr, col, n, n, kernels.DEGREE_BOTH)
            counts = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
            weight = torch.repeat_interleave(rowscale[:n], counts) * colscale[col.to(torch.int64)]
        g = CSRGraph(rowptr, col, weight, n, n, symmetric=True)
        g.ndata = {"feat": feat.t().contiguous()}
        g.edata = {"weight": weight}
        data.data.uns[self.out] = g
        return data
