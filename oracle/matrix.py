"""Oracle for dance/utils/matrix.py (test infrastructure; see oracle/__init__.py).

Pinned by the reference's known-answer tests (tests/utils/test_matrix.py:9-65, vectors in
tests/golden/matrix_known_answers.json).  numba typing notes are [3P-memory] (numba 0.59 is not installable here).
"""
import numpy as np


def normalize(mat, *, mode="normalize", axis=0, eps=-1.0):
    """dance/utils/matrix.py:8-67 (numpy branch)."""
    if not isinstance(mat, np.ndarray):
        raise TypeError(f"Invalid type for input matrix: {type(mat)}")
    opts = {"axis": axis, "keepdims": True}
    if mode == "standardize":  # :32-38
        shift = -mat.mean(**opts)
    elif mode == "minmax":
        min_vals = mat.min(**opts)
        shift = -min_vals
    else:
        shift = 0
    if mode == "normalize":  # :41-52
        denom = mat.sum(**opts)
    elif mode == "standardize":
        denom = mat.std(**opts)
    elif mode == "minmax":
        denom = mat.max(**opts) - min_vals
    elif mode == "l2":
        denom = (mat**2).sum(**opts)**0.5
    else:
        denom = None
    if denom is None:  # :55-62
        denom = 1
    elif eps == -1:
        denom[denom == 0] = 1
    elif eps > 0:
        denom = denom + eps
    else:
        raise ValueError(f"Invalid {eps=!r}. Must be positive or -1, the later set zero entries to one.")
    return (mat + shift) / denom  # :64


def mean_rank_data(x):
    """matrix.py:119-140 — average ranks (scipy rankdata 'average'), float32."""
    x = np.asarray(x, dtype=np.float32)
    less = (x[None, :] < x[:, None]).sum(1)
    equal = (x[None, :] == x[:, None]).sum(1)
    return (less + (equal + 1) / 2).astype(np.float32)


def _euclidean(x):
    """matrix.py:100-105: `sum = 0` (int) += f32 squares  ->  float64 accumulator, f32 terms, one final rounding."""
    x = np.asarray(x, dtype=np.float32)
    n, d = x.shape
    acc = np.zeros((n, n), dtype=np.float64)
    for t in range(d):  # index order, like the reference loop
        diff = x[:, None, t] - x[None, :, t]  # f32
        acc += (diff * diff).astype(np.float64)  # f32 square, f64 accumulate
    return np.sqrt(acc).astype(np.float32)


def _pearson_rows(x):
    """matrix.py:108-116, in the reference's mixed precision: f32 means/deviations/sums, the product of the two
    f32 sums raised to 0.5 and the quotient in double, return value rounded to f32."""
    x = np.asarray(x, dtype=np.float32)
    n, d = x.shape
    avg = np.empty(n, dtype=np.float32)
    for i in range(n):
        s = np.float32(0)
        for v in x[i]:
            s = np.float32(s + v)
        avg[i] = np.float32(s / np.float32(d))
    dev = (x - avg[:, None]).astype(np.float32)
    out = np.empty((n, n), dtype=np.float32)
    ss = np.empty(n, dtype=np.float32)
    for i in range(n):
        s = np.float32(0)
        for v in dev[i]:
            s = np.float32(s + np.float32(v * v))
        ss[i] = s
    for i in range(n):
        for j in range(n):
            cov = np.float32(0)
            for t in range(d):
                cov = np.float32(cov + np.float32(dev[i, t] * dev[j, t]))
            sq = float(np.float32(ss[i] * ss[j]))**0.5
            with np.errstate(divide="ignore", invalid="ignore"):
                out[i, j] = np.float32(1 - np.float64(cov) / np.float64(sq))
    return out


def pairwise_distance(x, dist_func_id=0):
    """matrix.py:164-180 — all-pairs distance matrix, float32 [n, n]."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if dist_func_id == 0:
        return _euclidean(x)
    if dist_func_id == 1:
        return _pearson_rows(x)
    if dist_func_id == 2:  # matrix.py:144-157
        ranks = np.stack([mean_rank_data(r) for r in x])
        return _pearson_rows(ranks)
    raise ValueError("Unknown distance function ID")


def pairwise_distance_f64(x, dist_func_id=0):
    """Double-precision ground truth for tolerance checks (not a restatement)."""
    x = np.asarray(x, dtype=np.float64)
    if dist_func_id == 0:
        return np.sqrt(((x[:, None, :] - x[None, :, :])**2).sum(-1))
    if dist_func_id == 2:
        x = np.stack([mean_rank_data(r) for r in x]).astype(np.float64)
    c = x - x.mean(1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        return 1 - (c @ c.T) / np.sqrt(np.outer((c * c).sum(1), (c * c).sum(1)))
