"""Exact PCA scores on the device ("GPU PCA") for the feature transforms in front of the graph builders
(dance/transforms/cell_feature.py:49-75 WeightedFeaturePCA, :160-185 CellPCA, which call sklearn ``PCA`` on the host).

The heavy products run on dh_gemm_f32 (exact-fp32 matrix cores): the Gram / covariance matrix of the centred data over
its LONG dimension (a K = n_cells reduction — the same shape as the layer's dW) and the projection.  The small symmetric
eigenproblem (at most n_genes x n_genes) is ``torch.linalg.eigh``.  Signs follow scikit-learn 1.3's ``svd_flip`` (the
reference's pin: largest-magnitude entry of every left singular vector positive).

This is the deterministic full decomposition; sklearn's ``svd_solver="auto"`` switches to a randomised SVD for these
shapes, so the host path is only reproduced up to that solver's approximation (and, without ``random_state``, not even by
itself run to run) — the device path is therefore opt-in (``device=`` on the transforms), never a silent substitution.
"""
from typing import Tuple

import torch

from .. import kernels


def pca_scores(X: torch.Tensor, n_components: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``sklearn.decomposition.PCA(n_components, svd_solver="full").fit_transform(X)`` for a device matrix X [samples,
    features] fp32: returns (scores [samples, k], components [k, features] or None, explained_variance [k]).

    samples <= features: eigen-decomposition of the sample Gram matrix Xc Xc^T, scores = U S (components are not formed).
    samples  > features: eigen-decomposition of Xc^T Xc, components = V^T, scores = Xc V."""
    if X.dim() != 2 or X.dtype != torch.float32:
        raise TypeError("pca_scores expects a 2-d float32 device tensor")
    n, f = X.shape
    k = int(n_components)
    if not 0 < k <= min(n, f):
        raise ValueError(f"n_components={k} must be between 1 and min(n_samples, n_features)={min(n, f)}")
    xc = X - X.mean(0, keepdim=True)  # centre every feature (column), as sklearn does
    if n <= f:
        gram = kernels.gemm(xc, xc, trans_b=True)                    # [n, n], reduction over the features
        lam, u = torch.linalg.eigh(gram.double())
        lam, u = lam.flip(0)[:k].clamp(min=0), u.flip(1)[:, :k]
        signs = torch.sign(u.gather(0, u.abs().argmax(0, keepdim=True))).reshape(1, -1)
        signs[signs == 0] = 1
        scores = (u * signs * lam.sqrt().reshape(1, -1)).float()
        comps = None
    else:
        cov = kernels.gemm(xc, xc, trans_a=True)                     # [f, f], reduction over the samples
        lam, v = torch.linalg.eigh(cov.double())
        lam, v = lam.flip(0)[:k].clamp(min=0), v.flip(1)[:, :k].float().contiguous()
        scores = kernels.gemm(xc, v)                                 # [n, k]
        signs = torch.sign(scores.gather(0, scores.abs().argmax(0, keepdim=True))).reshape(1, -1)  # == sign of U's entry
        signs[signs == 0] = 1
        scores = scores * signs
        comps = (v * signs).t().contiguous()
    return scores, comps, (lam / max(n - 1, 1)).float()
