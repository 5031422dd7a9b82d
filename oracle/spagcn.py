"""Oracle for SpaGCN's module-level numerics (test infrastructure; see oracle/__init__.py).
Restates dance/modules/spatial/spatial_domain/spagcn.py:249-287 (calculate_p, search_l) and :807-809
(calc_adj_exp) in numpy exactly as written; pinned indirectly: SimpleGCDEC itself is pinned by
tests/golden/model_heads.npz (reference code run by AST extraction)."""
import numpy as np


def calc_adj_exp(adj, l):
    return np.exp(-1 * (adj**2) / (2 * (l**2)))  # spagcn.py:808


def calculate_p(adj, l):
    return np.mean(np.sum(calc_adj_exp(adj, l), 1)) - 1  # spagcn.py:249-251


def search_l(p, adj, start=0.01, end=1000, tol=0.01, max_run=100):
    """spagcn.py:254-287."""
    run = 0
    p_low = calculate_p(adj, start)
    p_high = calculate_p(adj, end)
    if p_low > p + tol or p_high < p - tol:
        return None
    elif np.abs(p_low - p) <= tol:
        return start
    elif np.abs(p_high - p) <= tol:
        return end
    while (p_low + tol) < p < (p_high - tol):
        run += 1
        if run > max_run:
            return None
        mid = (start + end) / 2
        p_mid = calculate_p(adj, mid)
        if np.abs(p_mid - p) <= tol:
            return mid
        if p_mid <= p:
            start, p_low = mid, p_mid
        else:
            end, p_high = mid, p_mid
    return None
