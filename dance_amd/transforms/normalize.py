"""Count-matrix normalisation in front of the graph builders, on the device.

The reference reaches these through scanpy — ``NormalizeTotal`` / ``Log1P`` / ``NormalizeTotalLog1P``
(dance/transforms/normalize.py:531-679) and ``AnnDataTransform(sc.pp.normalize_total | sc.pp.log1p | sc.pp.scale)`` in the
clustering pipelines (graphsc.py:113-128, scdsc.py:113-131, sctag.py:119-139); scanpy (pin: 1.10.1, requirements.txt:19) is a
third-party dependency that is not vendored, so the arithmetic below restates its published algorithm
(``scanpy/preprocessing/_normalization.py`` normalize_total, ``_simple.py`` log1p / scale):

  normalize_total : counts_i = sum_g X[i,g]; with exclude_highly_expressed the genes with X[i,g] > max_fraction * counts_i in
                    ANY cell are left out of the size factor; target = target_sum or median(counts_i | counts_i > 0);
                    X[i,:] /= (counts_i or 1) / target
  log1p           : X = log(1 + X) [/ log(base)]
  scale           : per gene mean and unbiased variance in float64, std 0 -> 1, X = (X - mean) / std, clipped to +-max_value
                    (upper clip only without zero_center)

All passes run as HIP kernels (dh_rowsum_masked_f32, dh_col_any_gt_f32, dh_rowscale_log1p_f32, dh_col_moments_f32,
dh_col_standardize_f32).  The result STAYS ON THE DEVICE: the slot (``X`` / layer / obsm) receives a ``dance_amd.data.DeviceArray``,
which the next device transform reads without a copy and which turns into a numpy array only if host code asks for one — a chain
of these transforms in front of the device PCA and the graph builders does no host round trip between its steps.  A host matrix
(numpy / scipy) found in the slot is uploaded once, by the first transform of the chain.
"""
from typing import Optional

import numpy as np
import scipy.sparse as sp
import torch

from .. import kernels
from ..data import DeviceArray, to_device_matrix
from ..registry import register_preprocessor
from ..utils.matrix import normalize
from .base import BaseTransform


def _upload(x, device) -> torch.Tensor:
    """Device fp32 matrix of a slot value.  A DeviceArray is cloned (the kernels below run in place and the old slot value must
    stay what it was); host values are uploaded (sparse ones densified, as the reference does at normalize.py:622-625)."""
    if isinstance(x, DeviceArray):
        return x.tensor.to(device=device, dtype=torch.float32).clone(memory_format=torch.contiguous_format)
    return to_device_matrix(x, device).contiguous()


def normalize_total(X: torch.Tensor, target_sum: Optional[float] = None, *, exclude_highly_expressed: bool = False,
                    max_fraction: float = 0.05, inplace: bool = False):
    """scanpy.pp.normalize_total on a device matrix [cells, genes] fp32; returns (X_normalised, counts_per_cell)."""
    counts = kernels.rowsum_masked(X)
    if exclude_highly_expressed:
        hi = kernels.col_any_gt(X, counts * float(max_fraction))
        counts = kernels.rowsum_masked(X, (hi == 0).to(torch.uint8))
    if target_sum is None:
        pos = torch.sort(counts[counts > 0]).values
        if pos.numel() == 0:
            after = torch.full((), float("nan"), device=X.device)
        else:
            after = (pos[(pos.numel() - 1) // 2] + pos[pos.numel() // 2]) / 2  # numpy's median
    else:
        after = torch.tensor(float(target_sum), dtype=torch.float32, device=X.device)
    div = (counts + (counts == 0).float()) / after
    return kernels.rowscale_log1p(X, div, log1p=False, inplace=inplace), counts


def log1p(X: torch.Tensor, base: Optional[float] = None, *, inplace: bool = False) -> torch.Tensor:
    return kernels.rowscale_log1p(X, None, log1p=True, base=base, inplace=inplace)


def scale(X: torch.Tensor, zero_center: bool = True, max_value: Optional[float] = None, *, inplace: bool = False):
    """scanpy.pp.scale on a device matrix; returns (X_scaled, mean [genes] f64, std [genes] f64)."""
    n = X.shape[0]
    s, q = kernels.col_moments(X)
    mean = s / n
    var = (q / n - mean * mean) * (n / max(n - 1, 1))
    std = var.sqrt()
    std_div = torch.where(std == 0, torch.ones_like(std), std)
    out = kernels.col_standardize(X, mean if zero_center else None, std_div, max_value, inplace=inplace)
    return out, mean, std


class _DeviceMatrixTransform(BaseTransform):
    """Shared plumbing: pick the matrix (X / layer / obsm), run on ``device``, write back."""

    def __init__(self, *, layer: Optional[str] = None, obsm: Optional[str] = None, device: str = "cuda", **kwargs):
        super().__init__(**kwargs)
        self.layer, self.obsm, self.device = layer, obsm, device

    def _read(self, data):
        ad = data.data
        x = ad.layers[self.layer] if self.layer is not None else ad.obsm[self.obsm] if self.obsm is not None else ad.X
        return _upload(x, self.device)

    def _write(self, data, x: torch.Tensor):
        arr = DeviceArray(x)  # stays on the device; numpy on demand
        ad = data.data
        if self.layer is not None:
            ad.layers[self.layer] = arr
        elif self.obsm is not None:
            ad.obsm[self.obsm] = arr
        else:
            ad.X = arr


@register_preprocessor("normalize")
class Log1P(_DeviceMatrixTransform):
    """``X = log(X + 1)`` (natural unless ``base``): dance/transforms/normalize.py:531-569."""

    _DISPLAY_ATTRS = ("base", "layer", "obsm")

    def __init__(self, base: Optional[float] = None, copy: bool = False, chunked: Optional[bool] = None, chunk_size: Optional[int] = None,
                 layer: Optional[str] = None, obsm: Optional[str] = None, **kwargs):
        super().__init__(layer=layer, obsm=obsm, **kwargs)
        self.base = base

    def __call__(self, data):
        self._write(data, log1p(self._read(data), self.base, inplace=True))
        data.data.uns["log1p"] = {"base": self.base}  # scanpy records it
        return data


@register_preprocessor("normalize")
class NormalizeTotal(_DeviceMatrixTransform):
    """Counts-per-cell normalisation, highly expressed genes excluded from the size factor (normalize.py:572-626)."""

    _DISPLAY_ATTRS = ("target_sum", "max_fraction", "key_added", "layer")

    def __init__(self, target_sum: Optional[float] = None, max_fraction: float = 0.05, key_added: Optional[str] = None,
                 layer: Optional[str] = None, **kwargs):
        super().__init__(layer=layer, **kwargs)
        self.target_sum, self.max_fraction, self.key_added = target_sum, max_fraction, key_added
        if max_fraction == 1.0:
            self.logger.info("max_fraction set to 1.0, this is equivalent to setting exclude_highly_expressed=False.")

    def __call__(self, data):
        x, counts = normalize_total(self._read(data), self.target_sum, exclude_highly_expressed=True, max_fraction=self.max_fraction,
                                    inplace=True)
        self._write(data, x)
        if self.key_added is not None:
            data.data.obs[self.key_added] = counts.cpu().numpy()
        return data


@register_preprocessor("normalize")
class UpdateSizeFactors(BaseTransform):
    """``obs["n_counts"]`` = row sums of X, ``obs["size_factors"]`` = n_counts / median (dance/transforms/normalize.py:647-659).  The sums
    are taken on the device when X lives there (float64 accumulation; only the n_obs sums come to the host frame)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)

    def __call__(self, data):
        x = data.data.X
        if isinstance(x, DeviceArray):
            n_counts = x.tensor.sum(dim=1, dtype=torch.float64).to(x.tensor.dtype).cpu().numpy()
        elif sp.issparse(x):
            n_counts = np.asarray(x.sum(axis=1)).ravel()
        else:
            n_counts = np.asarray(x).sum(axis=1)
        data.data.obs["n_counts"] = n_counts
        data.data.obs["size_factors"] = data.data.obs.n_counts / np.median(data.data.obs.n_counts)
        return data


@register_preprocessor("normalize")
class NormalizeTotalLog1P(BaseTransform):
    """NormalizeTotal then Log1P in one upload (normalize.py:662-679)."""

    _DISPLAY_ATTRS = ("base", "target_sum", "max_fraction")

    def __init__(self, base=None, target_sum=None, max_fraction=0.05, *, device: str = "cuda", **kwargs):
        super().__init__(**kwargs)
        self.base, self.target_sum, self.max_fraction, self.device = base, target_sum, max_fraction, device

    def __call__(self, data):
        x = _upload(data.data.X, self.device)
        x, _ = normalize_total(x, self.target_sum, exclude_highly_expressed=True, max_fraction=self.max_fraction, inplace=True)
        data.data.X = DeviceArray(log1p(x, self.base, inplace=True))
        data.data.uns["log1p"] = {"base": self.base}
        return data


@register_preprocessor("normalize")
class Scale(_DeviceMatrixTransform):
    """``sc.pp.scale`` as the clustering pipelines call it through AnnDataTransform (scdsc.py:126, sctag.py:135): unit variance
    and (optionally) zero mean per gene; ``var['mean']`` / ``var['std']`` recorded like scanpy."""

    _DISPLAY_ATTRS = ("zero_center", "max_value", "layer", "obsm")

    def __init__(self, zero_center: bool = True, max_value: Optional[float] = None, layer: Optional[str] = None, obsm: Optional[str] = None,
                 **kwargs):
        super().__init__(layer=layer, obsm=obsm, **kwargs)
        self.zero_center, self.max_value = zero_center, max_value

    def __call__(self, data):
        x, mean, std = scale(self._read(data), self.zero_center, self.max_value, inplace=True)
        self._write(data, x)
        if self.obsm is None:
            data.data.var["mean"] = mean.cpu().numpy()
            data.data.var["std"] = std.cpu().numpy()
        return data


@register_preprocessor("normalize")
class ColumnSumNormalize(BaseTransform):
    """Split- / batch-wise ``dance.utils.matrix.normalize`` of X (normalize.py:26-101)."""

    _DISPLAY_ATTRS = ("axis", "mode", "eps", "split_names", "batch_key")

    def __init__(self, *, axis: int = 0, split_names=None, batch_key: Optional[str] = None, mode: str = "normalize", eps: float = -1,
                 **kwargs):
        super().__init__(**kwargs)
        self.axis, self.split_names, self.batch_key, self.mode, self.eps = axis, split_names, batch_key, mode, eps

    def _groups(self, data):
        if self.batch_key is not None:
            if self.split_names is not None:
                raise ValueError("Exactly one of split_names and batch_key can be specified, got: "
                                 f"split_names={self.split_names!r}, batch_key={self.batch_key!r}")
            cols = data.data.obs.columns.tolist()
            if self.batch_key not in cols:
                raise KeyError(f"batch_key={self.batch_key!r} not found in `.obs`. Available columns are: {cols}")
            col = data.data.obs[self.batch_key].to_numpy()
            return {f"batch:{b}": np.where(col == b)[0] for b in dict.fromkeys(col.tolist())}
        if self.split_names is None:
            return {"full": np.arange(data.shape[0])}
        if isinstance(self.split_names, str) and self.split_names == "ALL":
            return {f"split:{k}": np.asarray(v) for k, v in data._split_idx_dict.items()}
        if isinstance(self.split_names, list):
            return {f"split:{k}": np.asarray(data.get_split_idx(k)) for k in self.split_names}
        raise TypeError(f"Unsupported type {type(self.split_names)} for split_names: {self.split_names!r}")

    def __call__(self, data):
        if sp.issparse(data.data.X):
            self.logger.warning("Native support for sparse matrix is not implemented yet, converting to dense array explicitly.")
            data.data.X = data.data.X.toarray()
        for name, idx in self._groups(data).items():
            self.logger.info(f"Scaling {name} (n={len(idx):,})")
            data.data.X[idx] = normalize(data.data.X[idx], mode=self.mode, axis=self.axis, eps=self.eps)
        return data
