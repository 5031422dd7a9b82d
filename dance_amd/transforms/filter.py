"""Gene / cell filtering and highly-variable-gene selection in front of the graph builders, on the device: the
steps the clustering pipelines run through scanpy before they normalise (graphsc.py:111-119, scdsc.py:113-121) —
``sc.pp.filter_genes`` / ``sc.pp.filter_cells`` (here ``FilterGenesScanpy`` / ``FilterCellsScanpy``, dance/transforms/filter.py:55-280)
and ``sc.pp.highly_variable_genes`` with the dispersion flavours (``HighlyVariableGenesLogarithmizedByTopGenes`` /
``...ByMeanAndDisp``, filter.py:1219-1372) or, on counts, the ``seurat_v3`` flavour (``HighlyVariableGenesRawCount``, filter.py:1142-1192,
STAGATE's default: stagate.py:159-162) with the loess trend fitted here.

The per-gene / per-cell statistics (sums, counts of expressing cells, means and variances) are reductions over the N x G matrix and
run on the device, on a ``DeviceArray`` slot without a host copy; what is left — quantile bins, medians and a top-k over the G-long
statistic vectors — is host numpy / pandas on kilobytes.  Subsetting keeps the matrix on the device (index_select).

scanpy (pin 1.10.1) is not vendored and not installable here: the selection rules restate its published algorithm
(scanpy/preprocessing/_simple.py filter_genes / filter_cells, _highly_variable_genes.py _highly_variable_genes_single_batch) —
parity unpinned by reference output, pinned to oracle/normalize.py's numpy restatement and hand-computed cases.
"""
import warnings
from typing import List, Optional, Union

import numpy as np
import torch

from ..data import DeviceArray, to_device_matrix
from ..registry import register_preprocessor
from .base import BaseTransform


def get_count(count_or_ratio: Optional[Union[float, int]], total: int) -> Optional[int]:
    """A count, or a ratio of ``total`` turned into one (filter.py:28-52)."""
    if count_or_ratio is None:
        return None
    if isinstance(count_or_ratio, float):
        if count_or_ratio > 1.:
            raise ValueError(f"{count_or_ratio=} is greater than 1. Ratio cannot be greater than 1.")
        return int(count_or_ratio * total)
    if isinstance(count_or_ratio, int):
        if count_or_ratio > total:
            raise ValueError(f"{count_or_ratio=} is greater than {total=}")
        return count_or_ratio
    raise TypeError(f"count_or_ratio must be either float or int, got {type(count_or_ratio)}")


def filter_mask(x: torch.Tensor, axis: int, *, min_counts=None, min_number=None, max_counts=None, max_number=None):
    """scanpy's filter_genes (axis 0: per gene over cells) / filter_cells (axis 1): exactly one threshold; ``*_counts`` compare the
    sum of the values, ``*_number`` the number of non-zero... positive entries.  Returns (keep mask, the statistic) as host arrays."""
    given = [v is not None for v in (min_counts, min_number, max_counts, max_number)]
    if sum(given) != 1:
        raise ValueError("Only provide one of the optional parameters `min_counts`, `min_genes`/`min_cells`, `max_counts`, "
                         "`max_genes`/`max_cells` per call.")
    if min_counts is not None or max_counts is not None:
        stat = x.sum(dim=axis, dtype=torch.float64).to(torch.float32)  # float32 sums like numpy's on a float32 matrix, to fp32 rounding
    else:
        stat = (x > 0).sum(dim=axis)
    lo = min_counts if min_counts is not None else min_number
    hi = max_counts if max_counts is not None else max_number
    keep = stat >= lo if lo is not None else stat <= hi
    return keep.cpu().numpy(), stat.cpu().numpy()


class FilterScanpy(BaseTransform):
    """Scanpy filtering transformation with additional options (ratios instead of counts)."""

    _FILTER_TARGET = None

    def __init__(self, min_counts=None, min_genes_or_cells=None, max_counts=None, max_genes_or_cells=None, split_name: Optional[str] = None,
                 channel: Optional[str] = None, channel_type: Optional[str] = "X", key_n_counts: Optional[str] = None,
                 key_n_genes_or_cells: Optional[str] = None, inplace=True, device="cuda", **kwargs):
        super().__init__(**kwargs)
        self.min_counts, self.min_genes_or_cells = min_counts, min_genes_or_cells
        self.max_counts, self.max_genes_or_cells = max_counts, max_genes_or_cells
        self.split_name, self.channel, self.channel_type = split_name, channel, channel_type
        self.key_n_counts, self.key_n_genes_or_cells, self.inplace, self.device = key_n_counts, key_n_genes_or_cells, inplace, device
        if self._FILTER_TARGET is None:
            raise NotImplementedError("Use FilterCellsScanpy or FilterGenesScanpy instead")

    def prepCounts(self, x: torch.Tensor, axis: int):
        """Thresholds given as a ratio in (0, 1) are percentiles of the per-gene / per-cell sums (filter.py:147-163)."""
        is_ratio = lambda v: isinstance(v, float) and 0 < v < 1
        if not (is_ratio(self.min_counts) or is_ratio(self.max_counts)):
            return self.min_counts, self.max_counts
        n_counts = x.sum(dim=axis, dtype=torch.float64).cpu().numpy()
        if isinstance(self.min_counts, float) and 0 <= self.min_counts <= 1:
            return np.percentile(n_counts, self.min_counts * 100), None
        return None, np.percentile(n_counts, self.max_counts * 100)

    def __call__(self, data):
        x = data.get_feature(return_type=self.device, split_name=self.split_name, channel=self.channel, channel_type=self.channel_type)
        total_cells, total_features = x.shape
        genes = self._FILTER_TARGET == "genes"
        axis = 0 if genes else 1
        min_counts, max_counts = self.prepCounts(x, axis)
        basis = total_cells if genes else total_features
        keep, _ = filter_mask(x, axis, min_counts=min_counts, max_counts=max_counts, min_number=get_count(self.min_genes_or_cells, basis),
                              max_number=get_count(self.max_genes_or_cells, basis))
        frame = data.data.var if genes else data.data.obs
        if self.key_n_counts is not None:
            frame[self.key_n_counts] = x.sum(dim=axis, dtype=torch.float64).to(torch.float32).cpu().numpy()
            if self.key_n_genes_or_cells is not None:
                frame[self.key_n_genes_or_cells] = (x > 0).sum(dim=axis).cpu().numpy()
        if not keep.all():
            self.logger.info(f"Subsetting {self._FILTER_TARGET} ({int((~keep).sum()):,} removed) due to {self}")
            if self.inplace:
                if genes:
                    data.data._inplace_subset_var(keep)
                else:
                    data.filter_by_mask(keep)
            else:
                kept = x[:, torch.as_tensor(np.flatnonzero(keep), device=x.device)] if genes else None
                if genes:
                    data.data.obsm[self.out] = DeviceArray(kept.contiguous())
                else:  # the reference stores x[:, subset].T here as well (filter.py:143); cells and genes swapped, kept as written
                    data.data.varm[self.out] = DeviceArray(x[:, torch.as_tensor(np.flatnonzero(keep), device=x.device)].t().contiguous())
        return data


@register_preprocessor("filter", "cell")
class FilterCellsScanpy(FilterScanpy):
    _DISPLAY_ATTRS = ("min_counts", "min_genes", "max_counts", "max_genes", "split_name")
    _FILTER_TARGET = "cells"

    def __init__(self, min_counts=None, min_genes=None, max_counts=None, max_genes=None, split_name: Optional[str] = None,
                 channel: Optional[str] = None, channel_type: Optional[str] = "X", key_n_counts: Optional[str] = None,
                 key_n_genes: Optional[str] = None, inplace=True, **kwargs):
        super().__init__(min_counts, min_genes, max_counts, max_genes, split_name, channel, channel_type, key_n_counts, key_n_genes, inplace,
                         **kwargs)
        self.min_genes, self.max_genes = min_genes, max_genes


@register_preprocessor("filter", "gene")
class FilterGenesScanpy(FilterScanpy):
    _DISPLAY_ATTRS = ("min_counts", "min_cells", "max_counts", "max_cells", "split_name")
    _FILTER_TARGET = "genes"

    def __init__(self, min_counts=None, min_cells=None, max_counts=None, max_cells=None, split_name: Optional[str] = None,
                 channel: Optional[str] = None, channel_type: Optional[str] = "X", key_n_counts: Optional[str] = None,
                 key_n_cells: Optional[str] = None, inplace=True, **kwargs):
        super().__init__(min_counts, min_cells, max_counts, max_cells, split_name, channel, channel_type, key_n_counts, key_n_cells, inplace,
                         **kwargs)
        self.min_cells, self.max_cells = min_cells, max_cells


@register_preprocessor("filter", "gene")
class FilterGenesCommon(BaseTransform):
    """Keep the genes that are expressed (non-zero somewhere) in every batch, or in every named split (filter.py:319-383).  The
    per-group |x| column sums are device reductions; the kept genes come out sorted by NAME, as the reference's
    ``_inplace_subset_var(sorted(names))`` leaves them."""

    _DISPLAY_ATTRS = ("batch_key", "split_keys")

    def __init__(self, batch_key: Optional[str] = None, split_keys: Optional[List[str]] = None, **kwargs):
        self.device = kwargs.pop("device", "cuda")
        super().__init__(**kwargs)
        if (batch_key is not None) and (split_keys is not None):
            raise ValueError("Either batch_key or split_keys can be specified, but not both. "
                             f"Got {batch_key=!r}, {split_keys=!r}")
        elif (batch_key is None) and (split_keys is None):
            raise ValueError("Either one of batch_key or split_keys must be specified.")
        self.batch_key, self.split_keys = batch_key, split_keys

    def __call__(self, data):
        x = data.get_feature(return_type=self.device, channel_type="X")
        if self.batch_key is None:
            groups = {k: np.asarray(data.get_split_idx(k, error_on_miss=True), dtype=np.int64) for k in self.split_keys}
        else:
            groups = {k: np.flatnonzero((data.data.obs[self.batch_key] == k).values) for k in data.data.obs[self.batch_key].unique()}
        common = None
        for name, rows in groups.items():
            hit = x[torch.as_tensor(rows, device=x.device)].abs().sum(0, dtype=torch.float64) > 0
            self.logger.info(f"{int(hit.sum()):,} genes found in {name!r}")
            common = hit if common is None else common & hit
        names = sorted(data.data.var_names[common.cpu().numpy()])
        self.logger.info(f"Found {len(names):,} common genes out of {data.shape[1]:,} total genes.")
        data.data._inplace_subset_var(names)


def gene_summary(x: torch.Tensor, mode: str) -> np.ndarray:
    """The per-gene statistic of dance's ``FilterGenes`` family (filter.py:476-487): ``sum``, ``var`` (population variance,
    mean(x^2) - mean(x)^2), ``cv`` = std / mean, ``rv`` = var / mean (both with 0 where the mean is 0) — float64 column
    reductions on ``x``'s device, G numbers back to the host."""
    n = x.shape[0]
    s = x.sum(0, dtype=torch.float64)
    if mode == "sum":
        return s.cpu().numpy()
    mean = s / n
    if mode == "var":
        q = torch.zeros_like(s)
        step = max(1, (1 << 27) // max(x.shape[1], 1))
        for lo in range(0, n, step):
            c = x[lo:lo + step].double()
            q += (c * c).sum(0)
        return (q / n - mean * mean).cpu().numpy()
    if mode in ("cv", "rv"):
        q = torch.zeros_like(s)
        step = max(1, (1 << 27) // max(x.shape[1], 1))
        for lo in range(0, n, step):
            c = x[lo:lo + step].double() - mean[None, :]  # numpy's var: mean of squared deviations
            q += (c * c).sum(0)
        var = q / n
        out = (var.sqrt() if mode == "cv" else var) / mean
        return torch.nan_to_num(out, nan=0.0, posinf=0.0, neginf=0.0).cpu().numpy()
    raise ValueError(f"Unknown summarization mode {mode!r}, available options are ['cv', 'rv', 'sum', 'var']")


class FilterGenes(BaseTransform):
    """Filter genes on a per-gene summary of the expression matrix (filter.py:437-518); subclasses say which genes stay."""

    def __init__(self, *, mode: str = "sum", channel: Optional[str] = None, channel_type: Optional[str] = None,
                 whitelist_indicators: Optional[Union[str, List[str]]] = None, add_n_counts=True, add_n_cells=True, inplace=True,
                 device="cuda", **kwargs):
        super().__init__(**kwargs)
        if (channel is not None) and (channel_type != "layers"):
            raise ValueError(f"Only X layers is available for filtering genes, specified {channel_type=!r}")
        if mode not in (all_modes := ["cv", "rv", "sum", "var"]):
            raise ValueError(f"Unknown summarization mode {mode!r}, available options are {all_modes}")
        self.mode, self.channel, self.channel_type, self.whitelist_indicators = mode, channel, channel_type, whitelist_indicators
        self.add_n_counts, self.add_n_cells, self.inplace, self.device = add_n_counts, add_n_cells, inplace, device

    def _get_preserve_mask(self, gene_summary: np.ndarray) -> np.ndarray:
        raise NotImplementedError

    def __call__(self, data):
        kw = dict(channel=self.channel, channel_type="layers") if self.channel is not None else dict(channel_type="X")
        x = data.get_feature(return_type=self.device, **kw)
        if self.add_n_counts:
            data.data.var["n_counts"] = x.sum(0, dtype=torch.float64).to(torch.float32).cpu().numpy()
        if self.add_n_cells:
            data.data.var["n_cells"] = (x > 0).sum(0).cpu().numpy()
        summary = gene_summary(x, self.mode)
        mask = self._get_preserve_mask(summary)
        selected = sorted(data.data.var_names[mask])
        if self.whitelist_indicators is not None:  # genes flagged in any of these .var columns stay whatever their statistic
            columns = [self.whitelist_indicators] if isinstance(self.whitelist_indicators, str) else self.whitelist_indicators
            flags = data.data.var[columns]
            selected = sorted(set(selected) | set(flags[flags.max(1)].index.tolist()))
        data.data.uns["gene_summary"] = summary
        self.logger.info(f"{data.shape[1] - len(selected):,} genes removed")
        if self.inplace:
            data.data._inplace_subset_var(selected)  # (sic) by sorted name: the columns are reordered
        else:
            pos = torch.as_tensor(data.data.var_names.get_indexer(selected), device=x.device)
            data.data.obsm[self.out] = DeviceArray(x.index_select(1, pos).contiguous())


@register_preprocessor("filter", "gene")
class FilterGenesPercentile(FilterGenes):
    """Keep the genes whose summary lies between two percentiles of all summaries, ends included (filter.py:521-588)."""

    _DISPLAY_ATTRS = ("min_val", "max_val", "mode")

    def __init__(self, min_val: Optional[float] = 1, max_val: Optional[float] = 99, *, mode: str = "sum", channel: Optional[str] = None,
                 channel_type: Optional[str] = None, whitelist_indicators: Optional[Union[str, List[str]]] = None, add_n_counts=True,
                 add_n_cells=True, inplace=True, **kwargs):
        super().__init__(mode=mode, channel=channel, channel_type=channel_type, whitelist_indicators=whitelist_indicators,
                         add_n_counts=add_n_counts, add_n_cells=add_n_cells, inplace=inplace, **kwargs)
        self.min_val, self.max_val = min_val, max_val

    def _get_preserve_mask(self, gene_summary):
        lo, hi = np.percentile(gene_summary, self.min_val), np.percentile(gene_summary, self.max_val)
        return np.logical_and(gene_summary >= lo, gene_summary <= hi)


@register_preprocessor("filter", "gene")
class FilterGenesTopK(FilterGenes):
    """Keep the ``num_genes`` genes with the largest (``top``) or smallest summary (filter.py:590-662)."""

    _DISPLAY_ATTRS = ("num_genes", "top", "mode")

    def __init__(self, num_genes: int = 1000, top: bool = True, *, mode: str = "cv", channel: Optional[str] = None,
                 channel_type: Optional[str] = "X", whitelist_indicators: Optional[Union[str, List[str]]] = None, add_n_counts=False,
                 add_n_cells=False, inplace=True, **kwargs):
        super().__init__(mode=mode, channel=channel, channel_type=channel_type, whitelist_indicators=whitelist_indicators,
                         add_n_counts=add_n_counts, add_n_cells=add_n_cells, inplace=inplace, **kwargs)
        self.num_genes, self.top = num_genes, top

    def _get_preserve_mask(self, gene_summary):
        total = gene_summary.size
        if self.num_genes >= total:
            self.logger.warning(f"{self.num_genes=!r} > total number of genes: {total}")
            self.num_genes = total
        order = gene_summary.argsort()
        mask = np.zeros(total, dtype=bool)
        mask[order[-self.num_genes:] if self.top else order[:self.num_genes]] = True
        return mask


class _ScanpyOrder(BaseTransform):
    """One scanpy threshold after another, in the order given (filter.py:1048-1139, :1403-1473)."""

    _KEYS: tuple = ()
    _STEP = None

    def _setup(self, order, thresholds, step_kwargs):
        self.order = list(self._KEYS) if order is None else list(order)
        self.logger.info(f"Filter order: {self.order}")
        if not set(self.order).issubset(thresholds):
            raise KeyError(f"An order should be in {thresholds.keys()}")
        self.steps = {key: self._STEP(**{key: thresholds[key]}, **step_kwargs) for key in thresholds if key in self.order}
        for key in thresholds:
            if key not in self.order:
                self.logger.warning(f"{key} not in order,It makes no sense to set {key}")

    def __call__(self, data):
        for key in self.order:
            self.steps[key](data)


@register_preprocessor("filter", "gene")
class FilterGenesScanpyOrder(_ScanpyOrder):
    _KEYS = ("min_counts", "min_cells", "max_counts", "max_cells")
    _STEP = FilterGenesScanpy

    def __init__(self, order: Optional[List[str]] = None, min_counts=None, min_cells=None, max_counts=None, max_cells=None,
                 split_name: Optional[str] = None, channel: Optional[str] = None, channel_type: Optional[str] = "X", add_n_counts=True,
                 add_n_cells=True, inplace=True, params_dict=None, **kwargs):
        device = kwargs.pop("device", "cuda")
        super().__init__(**kwargs)
        self.add_n_counts, self.add_n_cells = add_n_counts, add_n_cells
        self._setup(order, dict(min_counts=min_counts, min_cells=min_cells, max_counts=max_counts, max_cells=max_cells),
                    dict(split_name=split_name, channel=channel, channel_type=channel_type, key_n_counts="n_counts" if add_n_counts else None,
                         key_n_cells="n_cells" if add_n_cells else None, inplace=inplace, device=device, **kwargs))
        self.filter_genes_order, self.geneScanpyOrderDict = self.order, self.steps


@register_preprocessor("filter", "cell")
class FilterCellsScanpyOrder(_ScanpyOrder):
    _KEYS = ("min_counts", "min_genes", "max_counts", "max_genes")
    _STEP = FilterCellsScanpy

    def __init__(self, order: Optional[List[str]] = None, min_counts=None, min_genes=None, max_counts=None, max_genes=None,
                 split_name: Optional[str] = None, channel: Optional[str] = None, channel_type: Optional[str] = "X", add_n_counts=True,
                 add_n_genes=True, inplace=True, **kwargs):
        device = kwargs.pop("device", "cuda")
        super().__init__(**kwargs)
        self.add_n_counts, self.add_n_genes = add_n_counts, add_n_genes
        self._setup(order, dict(min_counts=min_counts, min_genes=min_genes, max_counts=max_counts, max_genes=max_genes),
                    dict(split_name=split_name, channel=channel, channel_type=channel_type, key_n_counts="n_counts" if add_n_counts else None,
                         key_n_genes="n_genes" if add_n_genes else None, inplace=inplace, device=device, **kwargs))
        self.filter_cells_order, self.cellScanpyOrderDict = self.order, self.steps


@register_preprocessor("filter", "gene")
class FilterGenesMatch(BaseTransform):
    """Remove the genes whose names start / end with one of the given strings (dance/transforms/filter.py:386-435: ERCC spike-ins,
    mitochondrial genes in SpaGCN's pipeline).  Names live on the host; only the kept columns of a device matrix are gathered."""

    _DISPLAY_ATTRS = ("prefixes", "suffixes")

    def __init__(self, prefixes=None, suffixes=None, case_sensitive: bool = False, **kwargs):
        super().__init__(**kwargs)
        self.prefixes, self.suffixes, self.case_sensitive = prefixes or [], suffixes or [], case_sensitive
        if case_sensitive:  # (sic: with case_sensitive=True the reference upper-cases both sides, :413-424)
            self.prefixes = [i.upper() for i in self.prefixes]
            self.suffixes = [i.upper() for i in self.suffixes]

    def __call__(self, data):
        names = data.data.var.index.astype(str)
        ids = names.str.upper() if self.case_sensitive else names
        remove = np.zeros(len(names), dtype=bool)
        for kind, items in (("prefix", self.prefixes), ("suffix", self.suffixes)):
            for item in items:
                hit = np.asarray(ids.str.startswith(item) if kind == "prefix" else ids.str.endswith(item))
                self.logger.info(f"{int(hit.sum())} number of genes will be removed due to {kind} {item!r}")
                remove |= hit
        self.logger.info(f"Removing {int(remove.sum())} genes in total")
        if remove.any():
            data.data._inplace_subset_var(~remove)
        return data


@register_preprocessor("filter", "cell")
class FilterCellsType(BaseTransform):
    """Drop the cells of every cell type with at most ``cell_type_threshold`` cells (dance/transforms/filter.py:1477-1512; scHeteroNet's
    pipeline).  ``obsm["cell_type"]`` is the one-hot label frame of the reference's datasets."""

    def __init__(self, cell_type_threshold=10, **kwargs):
        super().__init__(**kwargs)
        self.cell_type_threshold = cell_type_threshold

    def __call__(self, data):
        import pandas as pd
        one_hot = data.data.obsm["cell_type"]
        if not isinstance(one_hot, pd.DataFrame):
            raise TypeError(f"Expected obsm['cell_type'] to be a pandas.DataFrame, but got {type(one_hot)}")
        counts = one_hot.sum(axis=0)
        rare = counts[counts <= self.cell_type_threshold].index
        self.logger.info(f"Found {len(rare)} cell types with counts <= {self.cell_type_threshold}: {rare.tolist()}")
        keep = np.ones(len(one_hot), dtype=bool) if rare.empty else ~np.asarray(one_hot[rare].sum(axis=1) > 0)
        data.filter_by_mask(keep)
        return data


def gene_mean_var(x: torch.Tensor, *, undo_log: bool, base: Optional[float] = None):
    """Per-gene mean and unbiased variance (float64 accumulation) of x, or of expm1(x) for the "seurat" flavour — in row chunks, so
    the un-logged matrix never exists as a whole."""
    n, g = x.shape
    s = torch.zeros(g, dtype=torch.float64, device=x.device)
    q = torch.zeros(g, dtype=torch.float64, device=x.device)
    step = max(1, (1 << 27) // max(g, 1))
    for lo in range(0, n, step):
        c = x[lo:lo + step]
        if undo_log:
            c = torch.expm1(c * float(np.log(base)) if base is not None else c)
        c = c.double()
        s += c.sum(0)
        q += (c * c).sum(0)
    mean = s / n
    var = (q / n - mean * mean) * (n / max(n - 1, 1))
    return mean.cpu().numpy(), var.cpu().numpy()


def dispersion_hvg(mean: np.ndarray, var: np.ndarray, *, flavor: str = "seurat", n_top_genes: Optional[int] = None, n_bins: int = 20,
                   min_mean: float = 0.0125, max_mean: float = 3, min_disp: float = 0.5, max_disp: float = np.inf):
    """scanpy's dispersion-based selection from per-gene (mean, variance): returns (highly_variable mask, means, dispersions,
    dispersions_norm) as scanpy writes them to ``.var``.  [3P-memory: scanpy 1.10.1 _highly_variable_genes_single_batch]"""
    import pandas as pd
    if flavor not in ("seurat", "cell_ranger"):
        raise ValueError('`flavor` needs to be "seurat" or "cell_ranger" (the dispersion-based flavours)')
    mean = mean.astype(np.float64).copy()
    mean[mean == 0] = 1e-12
    dispersion = var / mean
    if flavor == "seurat":
        dispersion[dispersion == 0] = np.nan
        dispersion = np.log(dispersion)
        mean = np.log1p(mean)
    df = pd.DataFrame({"means": mean, "dispersions": dispersion})
    if flavor == "seurat":
        df["mean_bin"] = pd.cut(df["means"], bins=n_bins)
        grouped = df.groupby("mean_bin", observed=False)["dispersions"]
        centre, spread = grouped.mean(), grouped.std(ddof=1)
        single = spread.isnull()  # one gene in the bin: its normalised dispersion becomes 1
        spread[single.values] = centre[single.values].values
        centre[single.values] = 0
    else:
        df["mean_bin"] = pd.cut(df["means"], np.r_[-np.inf, np.percentile(df["means"], np.arange(10, 105, 5)), np.inf])
        grouped = df.groupby("mean_bin", observed=False)["dispersions"]
        centre = grouped.median()
        with np.errstate(invalid="ignore"):  # statsmodels.robust.mad: median(|x - median(x)|) / 0.6744897501960817
            spread = grouped.apply(lambda v: np.median(np.abs(v - np.median(v))) / 0.6744897501960817 if len(v) else np.nan)
    with np.errstate(divide="ignore", invalid="ignore"):
        norm = (df["dispersions"].values - centre[df["mean_bin"].values].values) / spread[df["mean_bin"].values].values
    norm = norm.astype(np.float32)
    if n_top_genes is not None:
        ok = np.sort(norm[~np.isnan(norm)])[::-1]
        n_top = min(int(n_top_genes), len(mean))
        if n_top > ok.size:
            n_top = ok.size
        cut = ok[n_top - 1] if n_top > 0 else np.inf
        hv = np.nan_to_num(norm) >= cut
    else:
        z = norm.copy()
        z[np.isnan(z)] = 0
        hv = np.logical_and.reduce((mean > min_mean, mean < max_mean, z > min_disp, z < max_disp))
    return hv, mean, dispersion, norm


def loess_at_points(x: torch.Tensor, y: torch.Tensor, span: float = 0.3, degree: int = 2) -> torch.Tensor:
    """Cleveland's local regression of ``y`` on ``x`` evaluated at every ``x`` (gaussian family, tricube kernel) — the model
    ``skmisc.loess.loess(x, y, span=span, degree=degree)`` fits for scanpy's ``seurat_v3`` flavour.  For each point the
    ``floor(n * span + 1e-5)`` nearest points get the weight (1 - (d / rho)^3)^3, rho = the distance to the farthest of them,
    and a polynomial in (x - q) / rho is fitted by weighted least squares; the fitted value is its intercept.  Batched in
    float64 on ``x``'s device: a block of query points against all points at a time -> weighted moment sums; the small normal
    equations are then solved on the host.  This is loess's DIRECT surface; scikit-misc's default interpolates the same vertex fits through a kd tree."""
    x, y = x.double(), y.double()
    n = x.numel()
    nf = min(n, int(np.floor(n * span + 1e-5)))
    if nf < 1:
        raise ValueError(f"span={span} is too small for {n} points")
    out = torch.empty_like(x)
    block = max(1, (1 << 24) // max(n, 1))
    for lo in range(0, n, block):
        q = x[lo:lo + block]
        d = (x[None, :] - q[:, None]).abs()
        rho = torch.kthvalue(d, nf, dim=1).values * max(1.0, span)
        safe = torch.where(rho > 0, rho, torch.ones_like(rho))
        r = d / safe[:, None]
        w = torch.where(r < 1, (1 - r**3)**3, torch.zeros_like(r))
        w = torch.where((rho > 0)[:, None], w, (d == 0).double())  # the window is one value of x repeated: its plain mean
        u = (x[None, :] - q[:, None]) / safe[:, None]
        powers = [torch.ones_like(u)]
        for _ in range(2 * degree):
            powers.append(powers[-1] * u)
        mom = [(w * p).sum(1) for p in powers]
        rhs = torch.stack([(w * powers[k] * y[None, :]).sum(1) for k in range(degree + 1)], 1)
        gram = torch.stack([torch.stack([mom[a + b] for b in range(degree + 1)], 1) for a in range(degree + 1)], 1)
        # the (degree+1)^2 normal equations of each point are solved on the host (bytes per point; pseudo-inverse, so a window
        # with fewer distinct x than coefficients gets the minimum-norm fit, as loess's own QR does)
        inv = np.linalg.pinv(gram.cpu().numpy(), hermitian=True)
        out[lo:lo + block] = torch.from_numpy((inv @ rhs.cpu().numpy()[:, :, None])[:, 0, 0]).to(x.device)
    return out


def seurat_v3_hvg(x: torch.Tensor, *, n_top_genes: int, span: float = 0.3, batches=None):
    """scanpy's ``seurat_v3`` selection on a device count matrix [3P-memory: scanpy 1.10.1 _highly_variable_genes_seurat_v3]:
    per-gene mean / unbiased variance, loess of log10(variance) on log10(mean) over the non-constant genes, counts clipped at
    mean + sqrt(N) * fitted std, variance of the standardised clipped counts; the genes are ranked by it.  With ``batches`` (one
    label per cell) all of that runs inside every batch; a gene's rank is the median of its ranks among the batches that have it in
    their top ``n_top_genes``, ties go to the gene more batches selected, and ``variances_norm`` is the mean over the batches.
    Returns (highly_variable, means, variances, variances_norm, rank, n_batches) as scanpy writes them (means / variances over all
    cells)."""
    if batches is not None:
        import pandas as pd
        cats = pd.Categorical(np.asarray(batches))
        per = []
        for code in range(len(cats.categories)):
            rows = torch.as_tensor(np.flatnonzero(cats.codes == code), device=x.device)
            per.append(_seurat_v3_norm_var(x.index_select(0, rows), span)[2])
        per = np.stack(per)
        ranks = np.argsort(np.argsort(-per, axis=1, kind="stable"), axis=1, kind="stable").astype(np.float32)
        n_batches = (ranks < n_top_genes).sum(0)
        ranks[ranks >= n_top_genes] = np.nan
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)  # a gene no batch selected: all-NaN column
            rank = np.nanmedian(ranks, axis=0)
        order = np.lexsort((-n_batches, np.where(np.isnan(rank), np.inf, rank)))   # rank ascending (NaN last), then more batches first
        hv = np.zeros(x.shape[1], dtype=bool)
        hv[order[:n_top_genes]] = True
        mean_np, var_np = gene_mean_var(x, undo_log=False)
        return hv, mean_np, var_np, per.mean(0), rank.astype(np.float32), n_batches
    mean_np, var_np, norm_var = _seurat_v3_norm_var(x, span)
    rank = np.argsort(np.argsort(-norm_var, kind="stable"), kind="stable").astype(np.float32)
    hv = rank < n_top_genes
    rank[~hv] = np.nan
    return hv, mean_np, var_np, norm_var, rank, None


def _seurat_v3_norm_var(x: torch.Tensor, span: float):
    """(means, variances, variance of the clipped standardised counts) of one batch: two passes over the matrix in row chunks."""
    n, g = x.shape
    mean_np, var_np = gene_mean_var(x, undo_log=False)
    mean = torch.from_numpy(mean_np).to(x.device)
    var = torch.from_numpy(var_np).to(x.device)
    est = torch.zeros(g, dtype=torch.float64, device=x.device)
    ok = var > 0
    if ok.any():
        est[ok] = loess_at_points(torch.log10(mean[ok]), torch.log10(var[ok]), span=span, degree=2)
    reg_std = torch.sqrt(10**est)
    clip = reg_std * float(np.sqrt(n)) + mean
    s = torch.zeros(g, dtype=torch.float64, device=x.device)
    q = torch.zeros(g, dtype=torch.float64, device=x.device)
    step = max(1, (1 << 27) // max(g, 1))
    for lo in range(0, n, step):
        c = torch.minimum(x[lo:lo + step].double(), clip[None, :])
        s += c.sum(0)
        q += (c * c).sum(0)
    norm_var = (n * mean * mean + q - 2 * s * mean) / ((n - 1) * reg_std * reg_std)
    return mean_np, var_np, norm_var.cpu().numpy()


def dispersion_hvg_batched(x: torch.Tensor, batches, names, *, flavor: str, base: Optional[float], n_bins: int, **rule):
    """scanpy's ``batch_key`` mode of the dispersion flavours [3P-memory: scanpy 1.10.1 _highly_variable_genes_batched]: the
    single-batch selection inside every batch over the genes expressed there (the others count as zeros), then per gene the mean of
    the statistics over the batches and the number of batches that selected it; with ``n_top_genes`` the genes are ranked by that
    number, ties by the averaged normalised dispersion (then by name), otherwise the cut-offs apply to the averages.  Per-batch
    statistics are device reductions over the batch's rows.  Returns a dict of the ``.var`` columns scanpy writes."""
    import pandas as pd
    g = x.shape[1]
    cats = pd.Categorical(np.asarray(batches))
    stats = {k: [] for k in ("means", "dispersions", "dispersions_norm", "highly_variable")}
    for code in range(len(cats.categories)):
        rows = torch.as_tensor(np.flatnonzero(cats.codes == code), device=x.device)
        xb = x.index_select(0, rows)
        expressed = ((xb > 0).sum(0) >= 1).cpu().numpy()
        cols = torch.as_tensor(np.flatnonzero(expressed), device=x.device)
        mean, var = gene_mean_var(xb.index_select(1, cols), undo_log=flavor == "seurat", base=base)
        hv, means, disp, norm = dispersion_hvg(mean, var, flavor=flavor, n_bins=n_bins, **rule)
        for key, val in (("means", means), ("dispersions", disp), ("dispersions_norm", norm), ("highly_variable", hv)):
            full = np.zeros(g, dtype=np.float64)
            full[expressed] = val
            stats[key].append(full)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)  # a gene whose statistic is NaN in every batch stays NaN
        out = {k: np.nanmean(np.stack(stats[k]), 0) for k in ("means", "dispersions", "dispersions_norm")}
    nb = np.stack(stats["highly_variable"]).sum(0).astype(np.int64)
    out["highly_variable_nbatches"] = nb
    out["highly_variable_intersection"] = nb == len(cats.categories)
    norm = out["dispersions_norm"]
    if rule.get("n_top_genes") is not None:
        by_name = np.argsort(np.asarray(names, dtype=str), kind="stable")            # the order the per-gene aggregation leaves
        key_norm = np.where(np.isnan(norm[by_name]), -np.inf, norm[by_name])        # NaN last
        ranked = by_name[np.lexsort((-key_norm, -nb[by_name]))]
        hv = np.zeros(g, dtype=bool)
        hv[ranked[:int(rule["n_top_genes"])]] = True
    else:
        z = np.nan_to_num(norm)
        out["dispersions_norm"] = z
        hv = np.logical_and.reduce((out["means"] > rule["min_mean"], out["means"] < rule["max_mean"], z > rule["min_disp"], z < rule["max_disp"]))
    out["highly_variable"] = hv
    return out


class _HVGBase(BaseTransform):

    def __init__(self, channel, channel_type, subset, inplace, batch_key, device, **kwargs):
        super().__init__(**kwargs)
        self.channel, self.channel_type, self.subset, self.inplace, self.device = channel, channel_type, subset, inplace, device
        self.batch_key = batch_key
        self.logger.info("Expects logarithmized data")

    def _select(self, data, **rule):
        kw = dict(channel=self.channel, channel_type=self.channel_type) if self.channel_type == "layers" else dict(channel_type="X")
        x = data.get_feature(return_type=self.device, **kw)
        base = data.data.uns.get("log1p", {}).get("base") if isinstance(data.data.uns.get("log1p"), dict) else None
        if self.batch_key is not None:
            cols = dispersion_hvg_batched(x, data.data.obs[self.batch_key].values, data.data.var_names, flavor=self.flavor, base=base,
                                          n_bins=self.n_bins, **rule)
            hv = cols["highly_variable"]
        else:
            mean, var = gene_mean_var(x, undo_log=self.flavor == "seurat", base=base)
            hv, means, disp, norm = dispersion_hvg(mean, var, flavor=self.flavor, n_bins=self.n_bins, **rule)
            cols = dict(highly_variable=hv, means=means, dispersions=disp, dispersions_norm=norm)
        if self.inplace:
            for key, val in cols.items():
                data.data.var[key] = val
            data.data.uns["hvg"] = {"flavor": self.flavor}
        if self.subset:
            data.data._inplace_subset_var(hv)
        return data


@register_preprocessor("filter", "gene")
class HighlyVariableGenesLogarithmizedByTopGenes(_HVGBase):
    """``sc.pp.highly_variable_genes(n_top_genes=...)`` on logarithmized data (filter.py:1219-1268), dispersion flavours."""

    _DISPLAY_ATTRS = ("n_top_genes", "n_bins", "flavor", "subset")

    def __init__(self, channel: Optional[str] = None, channel_type: Optional[str] = None, n_top_genes: Optional[int] = 1000, n_bins: int = 20,
                 flavor: str = "seurat", subset: bool = True, inplace: bool = True, batch_key: Optional[str] = None, device="cuda", **kwargs):
        super().__init__(channel, channel_type, subset, inplace, batch_key, device, **kwargs)
        self.n_top_genes, self.n_bins, self.flavor = n_top_genes, n_bins, flavor

    def __call__(self, data):
        return self._select(data, n_top_genes=self.n_top_genes)


@register_preprocessor("filter", "gene")
class HighlyVariableGenesLogarithmizedByMeanAndDisp(_HVGBase):
    """``sc.pp.highly_variable_genes`` with mean / dispersion cut-offs (filter.py:1314-1372), dispersion flavours."""

    _DISPLAY_ATTRS = ("min_disp", "max_disp", "min_mean", "max_mean", "n_bins", "flavor", "subset")

    def __init__(self, channel: Optional[str] = None, channel_type: Optional[str] = None, min_disp: Optional[float] = 0.5,
                 max_disp: Optional[float] = np.inf, min_mean: Optional[float] = 0.0125, max_mean: Optional[float] = 3, n_bins: int = 20,
                 flavor: str = "seurat", subset: bool = True, inplace: bool = True, batch_key: Optional[str] = None, device="cuda", **kwargs):
        super().__init__(channel, channel_type, subset, inplace, batch_key, device, **kwargs)
        self.min_disp, self.max_disp, self.min_mean, self.max_mean, self.n_bins, self.flavor = min_disp, max_disp, min_mean, max_mean, n_bins, flavor

    def __call__(self, data):
        return self._select(data, min_mean=self.min_mean, max_mean=self.max_mean, min_disp=self.min_disp, max_disp=self.max_disp)


@register_preprocessor("filter", "gene")
class HighlyVariableGenesRawCount(BaseTransform):
    """``sc.pp.highly_variable_genes(flavor="seurat_v3")`` on a COUNT matrix (filter.py:1142-1192): the loess fit of the
    variance-mean trend is computed here (``loess_at_points``, the direct surface — scikit-misc is not needed)."""

    _DISPLAY_ATTRS = ("n_top_genes", "span", "subset")

    def __init__(self, channel: Optional[str] = None, channel_type: Optional[str] = None, n_top_genes: Optional[int] = 1000,
                 span: Optional[float] = 0.3, subset: bool = True, inplace: bool = True, batch_key: Optional[str] = None,
                 check_values: bool = True, device="cuda", **kwargs):
        super().__init__(**kwargs)
        if n_top_genes is None:
            raise ValueError("`n_top_genes` is mandatory if `flavor` is `seurat_v3`.")  # scanpy's own check
        self.channel, self.channel_type, self.n_top_genes, self.span = channel, channel_type, n_top_genes, span
        self.subset, self.inplace, self.check_values, self.device, self.batch_key = subset, inplace, check_values, device, batch_key
        self.logger.info("Expects count data")

    def __call__(self, data):
        if data.data.X.shape[1] == 0:
            raise ValueError("Gene dimension is 0")
        kw = dict(channel=self.channel, channel_type=self.channel_type) if self.channel_type == "layers" else dict(channel_type="X")
        x = data.get_feature(return_type=self.device, **kw)
        if self.check_values and not bool((x[:min(len(x), 4096)] % 1 == 0).all()):
            warnings.warn("`flavor='seurat_v3'` expects raw count data, but non-integers were found.", UserWarning)
        batches = None if self.batch_key is None else data.data.obs[self.batch_key].values
        hv, means, variances, norm_var, rank, n_batches = seurat_v3_hvg(x, n_top_genes=int(self.n_top_genes), span=self.span, batches=batches)
        if self.inplace:
            v = data.data.var
            v["highly_variable"], v["highly_variable_rank"], v["means"] = hv, rank, means
            v["variances"], v["variances_norm"] = variances, norm_var
            if n_batches is not None:
                v["highly_variable_nbatches"] = n_batches
            data.data.uns["hvg"] = {"flavor": "seurat_v3"}
        if self.subset:
            data.data._inplace_subset_var(hv)
        return data
