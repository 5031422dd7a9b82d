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


def filter_genes(X, min_counts=None, min_cells=None, max_counts=None, max_cells=None):
    """scanpy.pp.filter_genes(X, inplace=False): (gene_subset, number_per_gene); exactly one threshold."""
    if sum(v is not None for v in (min_counts, min_cells, max_counts, max_cells)) != 1:
        raise ValueError("Only provide one of the optional parameters")
    X = np.asarray(X)
    number = X.sum(0) if (min_counts is not None or max_counts is not None) else (X > 0).sum(0)
    lo = min_counts if min_counts is not None else min_cells
    hi = max_counts if max_counts is not None else max_cells
    return (number >= lo if lo is not None else number <= hi), number


def filter_cells(X, min_counts=None, min_genes=None, max_counts=None, max_genes=None):
    """scanpy.pp.filter_cells(X, inplace=False): (cell_subset, number_per_cell)."""
    s, n = filter_genes(np.asarray(X).T, min_counts, min_genes, max_counts, max_genes)
    return s, n


def highly_variable_genes(X, flavor="seurat", n_top_genes=None, n_bins=20, min_mean=0.0125, max_mean=3, min_disp=0.5, max_disp=np.inf):
    """scanpy.pp.highly_variable_genes on logarithmized X, dispersion flavours, single batch — a plain numpy loop per bin
    (no pandas), restating _highly_variable_genes_single_batch [3P-memory, scanpy 1.10.1].  Returns (highly_variable, means,
    dispersions, dispersions_norm)."""
    X = np.asarray(X, dtype=np.float64)
    if flavor == "seurat":
        X = np.expm1(X)
    n = X.shape[0]
    mean = X.mean(0)
    var = ((X * X).mean(0) - mean**2) * (n / (n - 1))
    mean[mean == 0] = 1e-12
    disp = var / mean
    if flavor == "seurat":
        disp[disp == 0] = np.nan
        disp = np.log(disp)
        mean = np.log1p(mean)
        lo, hi = mean.min(), mean.max()
        edges = np.linspace(lo, hi, n_bins + 1)
        edges[0] = lo - (hi - lo) * 0.001                # pandas.cut(bins=int): left edge pushed out by 0.1 % of the range
        which = np.clip(np.searchsorted(edges, mean, side="left") - 1, 0, n_bins - 1)   # right-closed intervals
    else:
        edges = np.r_[-np.inf, np.percentile(mean, np.arange(10, 105, 5)), np.inf]
        which = np.searchsorted(edges, mean, side="left") - 1
    norm = np.full(mean.shape, np.nan)
    for b in np.unique(which):
        sel = which == b
        d = disp[sel]
        if flavor == "seurat":
            c = np.nanmean(d) if np.isfinite(d).any() else np.nan
            # pandas std(ddof=1) skips NaN; a single (non-NaN) gene gives NaN -> replaced by (mean, 0)
            k = np.isfinite(d).sum()
            s = np.nanstd(d, ddof=1) if k > 1 else np.nan
            if np.isnan(s):
                s, c = c, 0.0
        else:
            c = np.median(d)
            s = np.median(np.abs(d - c)) / 0.6744897501960817
        with np.errstate(divide="ignore", invalid="ignore"):
            norm[sel] = (d - c) / s
    norm = norm.astype(np.float32)
    if n_top_genes is not None:
        ok = np.sort(norm[~np.isnan(norm)])[::-1]
        n_top = min(n_top_genes, len(mean), ok.size)
        hv = np.nan_to_num(norm) >= ok[n_top - 1]
    else:
        z = np.where(np.isnan(norm), 0, norm)
        hv = (mean > min_mean) & (mean < max_mean) & (z > min_disp) & (z < max_disp)
    return hv, mean, disp, norm


def loess_direct(x, y, span=0.3, degree=2):
    """Local regression of y on x evaluated AT every x (Cleveland's loess, gaussian family, tricube kernel, the direct
    surface): for each point q the ``floor(n * span + 1e-5)`` nearest points are weighted by (1 - (d / rho)^3)^3 with rho the
    distance to the farthest of them, and a degree-``degree`` polynomial in (x - q) is fitted by weighted least squares; the
    fitted value is its intercept.  One ``numpy.linalg.lstsq`` per point — the plain form of netlib dloess ``ehg127``
    (sqrt-weights on the rows, pseudo-inverse for rank-deficient windows).  scikit-misc's default surface is the kd-tree /
    blending INTERPOLATION of exactly these vertex fits, so its values are close to these, not identical — **parity unpinned**."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n = len(x)
    nf = min(n, int(np.floor(n * span + 1e-5)))
    out = np.empty(n)
    for i in range(n):
        d = np.abs(x - x[i])
        rho = np.partition(d, nf - 1)[nf - 1] * max(1.0, span)
        w = np.where(d < rho, (1 - (d / rho)**3)**3, 0.0) if rho > 0 else (d == 0).astype(np.float64)
        sel = w > 0
        u = (x[sel] - x[i]) / (rho if rho > 0 else 1.0)
        A = np.stack([u**p for p in range(degree + 1)], 1) * np.sqrt(w[sel])[:, None]
        out[i] = np.linalg.lstsq(A, y[sel] * np.sqrt(w[sel]), rcond=None)[0][0]
    return out


def highly_variable_genes_seurat_v3(X, n_top_genes=1000, span=0.3):
    """scanpy.pp.highly_variable_genes(flavor="seurat_v3") on a count matrix, single batch [3P-memory, scanpy 1.10.1
    _highly_variable_genes_seurat_v3]: loess of log10(variance) on log10(mean) over the non-constant genes, counts clipped at
    mean + sqrt(N) * fitted std, variance of the standardised clipped counts, the ``n_top_genes`` largest kept.  Returns
    (highly_variable, means, variances, variances_norm, rank) — rank is NaN outside the selection, as scanpy writes it."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    mean = X.mean(0)
    var = ((X * X).mean(0) - mean**2) * (n / (n - 1))
    est = np.zeros(X.shape[1])
    ok = var > 0
    est[ok] = loess_direct(np.log10(mean[ok]), np.log10(var[ok]), span=span, degree=2)
    reg_std = np.sqrt(10**est)
    clipped = np.minimum(X, (reg_std * np.sqrt(n) + mean)[None, :])
    norm_var = (n * mean**2 + (clipped**2).sum(0) - 2 * clipped.sum(0) * mean) / ((n - 1) * reg_std**2)
    rank = np.argsort(np.argsort(-norm_var, kind="stable"), kind="stable").astype(np.float32)
    hv = rank < n_top_genes
    rank[~hv] = np.nan
    return hv, mean, var, norm_var, rank


def highly_variable_genes_batched(X, batch, names, flavor="seurat", n_top_genes=None, n_bins=20, min_mean=0.0125, max_mean=3, min_disp=0.5,
                                  max_disp=np.inf):
    """scanpy.pp.highly_variable_genes(batch_key=...) for the dispersion flavours [3P-memory, scanpy 1.10.1
    _highly_variable_genes_batched], written gene by gene: every batch runs the single-batch rule on the genes it expresses, a gene it
    does not express contributes zeros; per gene the batch values are averaged (NaN skipped) and the selections counted.  Ranking for
    ``n_top_genes``: more batches first, then larger mean normalised dispersion (NaN last), then the gene name."""
    X = np.asarray(X)
    batch = np.asarray(batch)
    g = X.shape[1]
    per_batch = []
    for b in sorted(set(batch.tolist())):
        xb = X[batch == b]
        expressed = (xb > 0).sum(0) >= 1
        hv, mean, disp, norm = highly_variable_genes(xb[:, expressed], flavor=flavor, n_top_genes=n_top_genes, n_bins=n_bins, min_mean=min_mean,
                                                     max_mean=max_mean, min_disp=min_disp, max_disp=max_disp)
        rows = np.zeros((4, g))
        rows[:, expressed] = np.stack([mean, disp, norm.astype(np.float64), hv.astype(np.float64)])
        per_batch.append(rows)
    per_batch = np.stack(per_batch)                       # batches x 4 x genes
    out = {}
    for i, key in enumerate(("means", "dispersions", "dispersions_norm")):
        vals = per_batch[:, i]
        out[key] = np.array([np.mean(v[~np.isnan(v)]) if (~np.isnan(v)).any() else np.nan for v in vals.T])
    nb = per_batch[:, 3].sum(0).astype(np.int64)
    out["highly_variable_nbatches"] = nb
    out["highly_variable_intersection"] = nb == per_batch.shape[0]
    if n_top_genes is not None:
        order = sorted(range(g), key=lambda j: (-nb[j], np.isnan(out["dispersions_norm"][j]),
                                                -out["dispersions_norm"][j] if not np.isnan(out["dispersions_norm"][j]) else 0.0, str(names[j])))
        hv = np.zeros(g, dtype=bool)
        hv[order[:n_top_genes]] = True
    else:
        z = np.nan_to_num(out["dispersions_norm"])
        out["dispersions_norm"] = z
        hv = (out["means"] > min_mean) & (out["means"] < max_mean) & (z > min_disp) & (z < max_disp)
    out["highly_variable"] = hv
    return out


def highly_variable_genes_seurat_v3_batched(X, batch, n_top_genes=1000, span=0.3):
    """scanpy's seurat_v3 flavour with ``batch_key`` [3P-memory, scanpy 1.10.1]: the single-batch statistic of
    ``highly_variable_genes_seurat_v3`` inside every batch, ranks within the batch, a gene's rank = the median of its in-top ranks,
    ``highly_variable_nbatches`` = how many batches have it in their top; chosen: the ``n_top_genes`` smallest median ranks, ties to
    the gene more batches picked.  Gene by gene, no vectorised ranking.  Returns (highly_variable, variances_norm, rank, n_batches)."""
    X = np.asarray(X, dtype=np.float64)
    batch = np.asarray(batch)
    g = X.shape[1]
    norm_vars = [highly_variable_genes_seurat_v3(X[batch == b], n_top_genes=g, span=span)[3] for b in sorted(set(batch.tolist()))]
    in_top = [[] for _ in range(g)]
    for nv in norm_vars:
        order = sorted(range(g), key=lambda j: (-nv[j], j))
        for r, j in enumerate(order[:n_top_genes]):
            in_top[j].append(r)
    rank = np.array([np.median(r) if r else np.nan for r in in_top])
    n_batches = np.array([len(r) for r in in_top])
    chosen = sorted(range(g), key=lambda j: (np.isnan(rank[j]), rank[j] if not np.isnan(rank[j]) else 0.0, -n_batches[j], j))[:n_top_genes]
    hv = np.zeros(g, dtype=bool)
    hv[chosen] = True
    return hv, np.mean(norm_vars, axis=0), rank, n_batches
