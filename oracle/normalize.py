"""TEST INFRASTRUCTURE ONLY (never imported by dance_amd/).  numpy restatement of the scanpy preprocessing calls the
reference pipelines make through AnnDataTransform / NormalizeTotal / Log1P (dance/transforms/normalize.py:531-679,
scdsc.py:113-131, sctag.py:119-139, graphsc.py:113-128).

Third-party algorithm: scanpy==1.10.1 (requirements.txt:19) is not vendored under /root/reference and not installed here, so
these functions restate its published algorithm (scanpy/preprocessing/_normalization.py::normalize_total,
_simple.py::log1p / scale_array, _utils.py::_get_mean_var) — **parity unpinned** against a running scanpy; pinned only to the
hand-computed vectors in tests/test_oracle_normalize.py.
"""
import numpy as np


def normalize_total(X, target_sum=None, exclude_highly_expressed=False, max_fraction=0.05):
    X = np.array(X, dtype=np.float32)
    counts = X.sum(1)
    if exclude_highly_expressed:
        gene_subset = (X > counts[:, None] * max_fraction).sum(0) == 0
        counts = X[:, gene_subset].sum(1)
    cell_subset = counts > 0
    after = np.median(counts[cell_subset]) if target_sum is None else target_sum
    counts = counts + (counts == 0)
    counts = counts / after
    return (X / counts[:, None]).astype(np.float32)


def log1p(X, base=None):
    X = np.log1p(np.array(X, dtype=np.float32))
    if base is not None:
        X = X / np.log(base)
    return X.astype(np.float32)


def scale(X, zero_center=True, max_value=None):
    X = np.array(X, dtype=np.float32)
    n = X.shape[0]
    mean = X.mean(0, dtype=np.float64)
    mean_sq = np.multiply(X, X).mean(0, dtype=np.float64)
    var = (mean_sq - mean**2) * (n / (n - 1))
    std = np.sqrt(var)
    std[std == 0] = 1
    if zero_center:
        X -= mean
    X /= std
    if max_value is not None:
        if zero_center:
            X = np.clip(X, -max_value, max_value)
        else:
            X[X > max_value] = max_value
    return X, mean, np.sqrt(var)
