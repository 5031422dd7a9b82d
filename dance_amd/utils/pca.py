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


def pca_scores_randomized(X: torch.Tensor, n_components: int, random_state=None, *, n_oversamples: int = 10,
                          n_iter="auto", ops=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``sklearn.decomposition.PCA(n_components, svd_solver="randomized", random_state=...).fit_transform(X)`` — the solver
    scikit-learn's ``"auto"`` picks for the reference's calls (dance/transforms/cell_feature.py:60-62: genes x cells with
    400 components; :176-181 CellPCA) — for a device matrix X [samples, features] fp32.

    Restated from Halko et al. 2011 (alg. 4.3 + 5.1) as scikit-learn 1.3 runs it (third-party, not part of the reference
    tree): Gaussian test matrix drawn from ``numpy.random.RandomState(random_state).normal(size=(short side, k + 10))`` —
    the same stream sklearn consumes, so a seeded host run and this one start from the same matrix — ``n_iter`` power
    iterations (7 if k < 0.1 min(shape) else 4) on the matrix oriented tall, an orthonormal basis Q of its range, the SVD
    of the small matrix Q^T A, signs by ``svd_flip`` on the sample-side vectors.  Every product with the long dimension
    is dh_gemm_f32; the bases are re-orthonormalised by Householder QR (``torch.linalg.qr``) where sklearn uses a pivoted
    LU — both only change the basis of the same subspace, so the results agree to rounding, not bit for bit.

    Returns (scores [samples, k] = U S, components [k, features], explained_variance [k]).  For these shapes the exact
    decomposition (``pca_scores``) is cheaper on this hardware and has no approximation error; this solver exists so that a
    seeded reference run can be reproduced."""
    import numpy as np

    if ops is None:
        ops = kernels
    if X.dim() != 2 or X.dtype != torch.float32:
        raise TypeError("pca_scores_randomized expects a 2-d float32 tensor")
    n, f = X.shape
    k = int(n_components)
    if not 0 < k <= min(n, f):
        raise ValueError(f"n_components={k} must be between 1 and min(n_samples, n_features)={min(n, f)}")
    size = k + int(n_oversamples)
    if n_iter == "auto":
        n_iter = 7 if k < 0.1 * min(n, f) else 4
    xc = (X - X.mean(0, keepdim=True)).contiguous()
    rng = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    # sklearn works on A = xc if samples >= features else xc^T (tall); the transposes below are operand flags, not copies
    tall_is_x = n >= f
    short = f if tall_is_x else n
    q = torch.from_numpy(rng.normal(size=(short, size)).astype(np.float32)).to(X.device)

    def a_times(m):     # A @ m: [long, size]
        return ops.gemm(xc, m) if tall_is_x else ops.gemm(xc, m, trans_a=True)

    def at_times(m):    # A^T @ m: [short, size]
        return ops.gemm(xc, m, trans_a=True) if tall_is_x else ops.gemm(xc, m)

    def orth(m):
        return torch.linalg.qr(m, mode="reduced")[0].contiguous()

    for _ in range(int(n_iter)):
        q = orth(a_times(q))
        q = orth(at_times(q))
    q = orth(a_times(q))                                   # [long, size], orthonormal basis of the sampled range of A
    bt = at_times(q)                                       # B^T = A^T Q: [short, size]
    v, s, uh_t = torch.linalg.svd(bt.double(), full_matrices=False)   # B = Q^T A = uh s v^T
    uh = uh_t.t().float().contiguous()                     # [size, size]
    u = ops.gemm(q, uh)                                    # [long, size]: left singular vectors of A
    v = v.float()                                          # [short, size]: right singular vectors of A
    samp, feat = (u, v) if tall_is_x else (v, u)           # sample-side [n, size] and feature-side [f, size] vectors of X
    signs = torch.sign(samp.gather(0, samp.abs().argmax(0, keepdim=True))).reshape(1, -1)
    signs[signs == 0] = 1
    s = s[:k].float()
    scores = (samp[:, :k] * signs[:, :k] * s.reshape(1, -1)).contiguous()
    comps = (feat[:, :k] * signs[:, :k]).t().contiguous()
    return scores, comps, s * s / max(n - 1, 1)
