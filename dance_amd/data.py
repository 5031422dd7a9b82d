"""Minimal stand-ins for the objects every DANCE transform / method receives.

``anndata`` is not installable in this environment, so ``AnnDataLite`` provides the attribute surface the hot
path touches (``X, obs, var, obsm, varm, obsp, varp, layers, uns``) and ``Data`` mirrors the accessor API of
dance/data/base.py (splits :114-184, config :203-271, ``get_feature`` :415-475, ``get_x/get_y/get_train_data``
:845-888).  A real ``anndata.AnnData`` can be wrapped by ``Data`` just as well — only attribute access is used.
"""
import copy
import warnings
from pprint import pformat
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
import torch


class DeviceArray:
    """A matrix that LIVES ON THE DEVICE, stored in an AnnData slot (``X``, ``obsm[...]``, ``varm[...]``, ``layers[...]``).

    Device transforms write these instead of numpy arrays and read them back without a copy, so a chain
    ``NormalizeTotal -> Log1P -> Scale -> WeightedFeaturePCA(device=...) -> CellFeatureGraph / NeighborGraph`` moves nothing
    over PCIe between its steps.  Host code that reads the slot still works: ``np.asarray`` / indexing /
    ``toarray()`` materialise a numpy copy on first use (cached; ``host_copies`` counts them).  Read-only by convention: a
    transform replaces the slot, it does not write into the old array."""

    host_copies = 0  # class-wide count of device -> host materialisations (the zero-round-trip tests watch it)

    def __init__(self, tensor: torch.Tensor):
        if tensor.dim() != 2:
            raise ValueError("DeviceArray wraps a 2-d matrix")
        self.tensor = tensor
        self._host = None

    @property
    def shape(self):
        return tuple(self.tensor.shape)

    @property
    def ndim(self):
        return 2

    @property
    def dtype(self):
        return np.dtype(str(self.tensor.dtype).replace("torch.", ""))

    def __len__(self):
        return self.tensor.shape[0]

    def numpy(self) -> np.ndarray:
        if self._host is None:
            self._host = self.tensor.detach().cpu().numpy()
            DeviceArray.host_copies += 1
        return self._host

    toarray = numpy  # Data.get_feature densifies through .toarray()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __getitem__(self, idx):
        return self.numpy()[idx]

    @property
    def T(self):
        return self.numpy().T

    def astype(self, dtype, **kw):
        return self.numpy().astype(dtype, **kw)

    def __getattr__(self, name):  # anything else an ndarray offers (sum, max, mean, reshape ...): on the materialised copy
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.tensor.dtype}, device={self.tensor.device})"

    def __iter__(self):
        return iter(self.numpy())

    __hash__ = object.__hash__  # (identity; __eq__ below is element-wise, as for an ndarray)


def _forward_operators():
    """Arithmetic / comparison operators of host code that treats the slot as an ndarray (``x / x.sum(1, keepdims=True)``, ``x > 0``):
    evaluated on the materialised host copy — operators are looked up on the type, ``__getattr__`` never sees them."""
    import operator
    binary = ["add", "sub", "mul", "truediv", "floordiv", "mod", "pow", "matmul", "lt", "le", "gt", "ge", "eq", "ne", "and", "or", "xor"]
    for name in binary:
        op = getattr(operator, name if name not in ("and", "or") else name + "_")
        setattr(DeviceArray, f"__{name}__", (lambda op: lambda self, other: op(self.numpy(), np.asarray(other) if isinstance(other, DeviceArray) else other))(op))
        if name in ("add", "sub", "mul", "truediv", "floordiv", "mod", "pow", "matmul", "and", "or", "xor"):
            setattr(DeviceArray, f"__r{name}__", (lambda op: lambda self, other: op(other, self.numpy()))(op))
    for name, op in (("neg", operator.neg), ("pos", operator.pos), ("abs", operator.abs), ("invert", operator.invert)):
        setattr(DeviceArray, f"__{name}__", (lambda op: lambda self: op(self.numpy()))(op))


_forward_operators()


def _is_lazy_graph(x) -> bool:
    """graph.LazyScipyCSR, by name: ``hasattr`` would run DeviceArray.__getattr__ (= a host copy) and graph.py imports this module."""
    return type(x).__name__ == "LazyScipyCSR"


def to_device_matrix(x, device) -> torch.Tensor:
    """The fp32 device tensor of an AnnData slot value: a ``DeviceArray`` hands over its tensor (no copy when it already lives
    on ``device``); numpy / scipy / DataFrame values are uploaded (one H2D)."""
    if isinstance(x, DeviceArray):
        return x.tensor.to(device=device, dtype=torch.float32)
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32)
    if _is_lazy_graph(x):  # graph.LazyScipyCSR (an obsp slot written by an on-device graph transform): densified like any sparse slot
        x = x.materialize()
    if sp.issparse(x):
        x = x.toarray()
    elif hasattr(x, "to_numpy"):
        x = x.to_numpy()
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(device)


class AnnDataLite:
    """Attribute container with AnnData's channel names (no views, no backing file)."""

    def __init__(self, X, obs=None, var=None, *, obsm=None, varm=None, obsp=None, varp=None, layers=None, uns=None):
        import pandas as pd
        self.X = X
        n_obs, n_var = X.shape
        self.obs = obs if obs is not None else pd.DataFrame(index=[str(i) for i in range(n_obs)])
        self.var = var if var is not None else pd.DataFrame(index=[str(i) for i in range(n_var)])
        self.obsm, self.varm = dict(obsm or {}), dict(varm or {})
        self.obsp, self.varp = dict(obsp or {}), dict(varp or {})
        self.layers, self.uns = dict(layers or {}), dict(uns or {})
        self.raw = None  # set by SaveRaw: a frozen copy of X / var (AnnData.raw)

    @property
    def shape(self):
        return self.X.shape

    @property
    def n_obs(self):
        return self.X.shape[0]

    @property
    def n_vars(self):
        return self.X.shape[1]

    @property
    def obs_names(self):
        return self.obs.index

    @property
    def var_names(self):
        return self.var.index

    # ---- in-place subsetting (AnnData._inplace_subset_var / _inplace_subset_obs): every aligned slot follows ---------------
    @staticmethod
    def _take(v, idx, axis):
        if isinstance(v, DeviceArray):
            t = v.tensor
            i = torch.as_tensor(idx, device=t.device)
            return DeviceArray(t.index_select(axis, i).contiguous())
        if _is_lazy_graph(v):  # graph.LazyScipyCSR: subset the scipy matrix it stands for (the device graph is not re-indexed)
            v = v.materialize()
        if sp.issparse(v):
            return v.tocsr()[idx] if axis == 0 else v.tocsc()[:, idx].tocsr()
        if hasattr(v, "iloc"):
            return v.iloc[idx]
        v = np.asarray(v) if not isinstance(v, (np.ndarray, torch.Tensor)) else v
        return v[idx] if axis == 0 else v[:, idx]

    @staticmethod
    def _positions(index, sel):
        """A boolean mask -> the kept positions in order; a list of names -> their positions IN THE LIST'S ORDER (what
        ``AnnData[:, names]`` gives: dance's gene filters pass sorted names, so the columns come out sorted by name)."""
        sel = np.asarray(sel)
        if sel.dtype == bool:
            return np.flatnonzero(sel)
        if sel.dtype.kind in "iu":
            return sel.astype(np.int64)
        pos = index.get_indexer(sel)
        if (pos < 0).any():
            raise KeyError(f"names not in the index: {[str(v) for v in sel[pos < 0][:5]]}")
        return pos

    def _inplace_subset_var(self, mask):
        idx = self._positions(self.var.index, mask)
        self.X = self._take(self.X, idx, 1)
        self.var = self.var.iloc[idx]
        self.varm = {k: self._take(v, idx, 0) for k, v in self.varm.items()}
        self.varp = {k: self._take(self._take(v, idx, 0), idx, 1) for k, v in self.varp.items()}
        self.layers = {k: self._take(v, idx, 1) for k, v in self.layers.items()}

    def _inplace_subset_obs(self, mask):
        idx = self._positions(self.obs.index, mask)
        if self.raw is not None:  # AnnData.raw follows the observations (not the variables)
            self.raw.X = self._take(self.raw.X, idx, 0)
            self.raw.shape = tuple(self.raw.X.shape)
        self.X = self._take(self.X, idx, 0)
        self.obs = self.obs.iloc[idx]
        self.obsm = {k: self._take(v, idx, 0) for k, v in self.obsm.items()}
        self.obsp = {k: self._take(self._take(v, idx, 0), idx, 1) for k, v in self.obsp.items()}
        self.layers = {k: self._take(v, idx, 0) for k, v in self.layers.items()}
        for k in [k for k in self.uns if isinstance(k, str) and k.endswith(".hip")]:
            del self.uns[k]  # device graphs over the old set of cells (written next to obsp by the graph transforms) are stale now

    def copy(self):
        return copy.deepcopy(self)

    def __getitem__(self, idx):
        """Rows ``idx`` (positions or a boolean mask) as a new container — the copy ``AnnData[idx].copy()`` would give."""
        if isinstance(idx, tuple):  # adata[rows, columns]
            rows, cols = idx
            out = self if isinstance(rows, slice) and rows == slice(None) else self[rows]
            if isinstance(cols, slice) and cols == slice(None):
                return out
            out = out.copy() if out is self else out
            out._inplace_subset_var(cols)
            return out
        idx = self._positions(self.obs.index, idx)
        out = AnnDataLite(self._take(self.X, idx, 0), obs=self.obs.iloc[idx], var=self.var.copy(),
                          obsm={k: self._take(v, idx, 0) for k, v in self.obsm.items()}, varm=dict(self.varm),
                          obsp={k: self._take(self._take(v, idx, 0), idx, 1) for k, v in self.obsp.items()}, varp=dict(self.varp),
                          layers={k: self._take(v, idx, 0) for k, v in self.layers.items()},
                          uns={k: v for k, v in self.uns.items() if not (isinstance(k, str) and k.endswith(".hip"))})
        if self.raw is not None:
            out.raw = copy.copy(self.raw)
            out.raw.X = self._take(self.raw.X, idx, 0)
            out.raw.shape = tuple(out.raw.X.shape)
        return out

    def __repr__(self):
        return (f"AnnDataLite object with n_obs x n_vars = {self.n_obs} x {self.n_vars}\n"
                f"    obsm: {list(self.obsm)}\n    varm: {list(self.varm)}\n    obsp: {list(self.obsp)}\n"
                f"    uns: {list(self.uns)}")


def _stack_rows(parts):
    if any(isinstance(p, DeviceArray) for p in parts):
        dev = next(p.tensor.device for p in parts if isinstance(p, DeviceArray))
        return DeviceArray(torch.cat([to_device_matrix(p, dev) for p in parts], 0))
    parts = [p.materialize() if _is_lazy_graph(p) else p for p in parts]
    if any(sp.issparse(p) for p in parts):
        return sp.vstack([sp.csr_matrix(p) for p in parts], format="csr")
    return np.concatenate([np.asarray(p) for p in parts], 0)


def concat(adatas, *, join: str = "inner", label: Optional[str] = None, keys=None, index_unique: Optional[str] = None):
    """Stack containers along the cells, the slice of ``anndata.concat`` that ``Data.append`` uses (dance/data/base.py:552):
    variables are matched by name (``join="inner"``: those every part has, in the first part's order), ``X`` / ``layers`` /
    ``obsm`` slots every part holds are stacked, ``obs`` keeps the columns every part has; pairwise slots, ``varm`` and
    ``uns`` are dropped, as anndata does with its default ``merge=None`` / ``uns_merge=None``."""
    import pandas as pd
    adatas = list(adatas)
    if join != "inner":
        raise NotImplementedError(f"join={join!r}: only the inner join is modelled")
    names = adatas[0].var.index
    for a in adatas[1:]:
        names = names[names.isin(a.var.index)]

    def cols(a, v):
        pos = a.var.index.get_indexer(names)
        if len(pos) == a.n_vars and (pos == np.arange(a.n_vars)).all():
            return v
        return AnnDataLite._take(v, pos, 1)

    obs = pd.concat([a.obs for a in adatas], join="inner")
    if label is not None:
        keys = list(range(len(adatas))) if keys is None else list(keys)
        obs[label] = pd.Categorical(np.repeat([str(k) for k in keys], [a.n_obs for a in adatas]))
    if index_unique is not None:
        keys = list(range(len(adatas))) if keys is None else list(keys)
        obs.index = [f"{i}{index_unique}{k}" for a, k in zip(adatas, keys) for i in a.obs.index]
    out = AnnDataLite(_stack_rows([cols(a, a.X) for a in adatas]), obs=obs, var=pd.DataFrame(index=names))
    for k in adatas[0].obsm:
        if all(k in a.obsm for a in adatas):
            out.obsm[k] = _stack_rows([a.obsm[k] for a in adatas])
    for k in adatas[0].layers:
        if all(k in a.layers for a in adatas):
            out.layers[k] = _stack_rows([cols(a, a.layers[k]) for a in adatas])
    return out


def _ensure_iter(x):
    return x if isinstance(x, (list, tuple)) else [x]


class Data:
    _FEATURE_CONFIGS: List[str] = ["feature_mod", "feature_channel", "feature_channel_type"]
    _LABEL_CONFIGS: List[str] = ["label_mod", "label_channel", "label_channel_type"]
    _DATA_CHANNELS: List[str] = ["obs", "var", "obsm", "varm", "obsp", "varp", "layers", "uns"]

    def __init__(self, data, train_size: Optional[int] = None, val_size: int = 0, test_size: int = -1,
                 split_index_range_dict: Optional[Dict[str, Tuple[int, int]]] = None,
                 full_split_name: Optional[str] = None):
        self._data = data
        self._split_idx_dict: Dict[str, Sequence[int]] = {}
        if split_index_range_dict is not None and full_split_name is not None:
            raise ValueError("Only one of split_index_range_dict, full_split_name can be specified, but not both")
        if split_index_range_dict is not None:
            for name, rng in split_index_range_dict.items():
                if not isinstance(rng, tuple) or len(rng) != 2 or any(not isinstance(i, int) for i in rng):
                    raise TypeError(f"The split index range must be a two-tuple of int, got {rng!r} for key {name!r}")
                if rng[1] - rng[0] > 0:
                    self._split_idx_dict[name] = list(range(*rng))
        elif full_split_name is not None:
            self._split_idx_dict[full_split_name] = list(range(self.shape[0]))
        else:
            self._setup_splits_default(train_size, val_size, test_size)
        if "dance_config" not in self._data.uns:
            self._data.uns["dance_config"] = dict()

    def _setup_splits_default(self, train_size, val_size, test_size):
        if train_size is None:
            return
        if isinstance(train_size, str) and train_size.lower() == "all":
            train_size, val_size, test_size = -1, 0, 0
        elif any(not isinstance(i, (int, np.integer)) for i in (train_size, val_size, test_size)):
            raise TypeError("Split sizes must be of type int")
        sizes = np.array((train_size, val_size, test_size))
        if (sizes == -1).sum() > 1:
            raise ValueError("Only one split can be specified as -1")
        n = self.num_cells
        for name, size in zip(("train", "val", "test"), sizes):
            if size < -1:
                raise ValueError(f"{name} must be integer no less than -1, got {size!r}")
            if size > n:
                raise ValueError(f"{name}={size:,} exceeds total number of samples {n:,}")
        if (tot := sizes.clip(0).sum()) > n:
            raise ValueError(f"Total size {tot:,} exceeds total number of samples {n:,}")
        sizes[sizes == -1] = n - sizes.clip(0).sum()
        edges = np.concatenate(([0], sizes.cumsum()))
        for i, name in enumerate(("train", "val", "test")):
            if edges[i + 1] - edges[i] > 0:
                self._split_idx_dict[name] = list(range(edges[i], edges[i + 1]))

    def filter_by_mask(self, mask, update_splits: bool = True):
        """Keep the cells where ``mask`` is True (dance/data/base.py:694-790); split indices are remapped to the new positions."""
        mask = np.asarray(mask, dtype=bool)
        if mask.ndim != 1 or len(mask) != self.shape[0]:
            raise ValueError(f"Mask length ({len(mask)}) must match number of cells ({self.shape[0]})")
        if mask.all():
            return self
        self._data._inplace_subset_obs(mask)
        if update_splits:
            new_pos = np.cumsum(mask) - 1
            # every split comes back SORTED by new position (base.py:780), whatever order it was given in; emptied splits stay as []
            self._split_idx_dict = {k: sorted(int(new_pos[i]) for i in v if mask[i]) for k, v in self._split_idx_dict.items()}
        return self

    # ---- basic views ---------------------------------------------------------------------------------------
    @property
    def data(self):
        return self._data

    @property
    def shape(self):
        return self._data.shape

    @property
    def num_cells(self) -> int:
        return self._data.shape[0]

    @property
    def num_features(self) -> int:
        return self._data.shape[1]

    @property
    def cells(self) -> List[str]:
        return self._data.obs.index.tolist()

    @property
    def x(self):
        return self.get_x(return_type="default")

    @property
    def y(self):
        return self.get_y(return_type="default")

    def __getitem__(self, idx):
        return self._data[idx]

    def copy(self):
        return copy.deepcopy(self)

    @property
    def config(self) -> Dict[str, Any]:
        return self._data.uns["dance_config"]

    def __getattr__(self, name):  # pass X / obs / obsm ... through, like the reference does with setattr
        if name in Data._DATA_CHANNELS + ["X"] and "_data" in self.__dict__:
            return getattr(self._data, name)
        raise AttributeError(name)

    def __repr__(self):
        return f"{self.__class__.__name__} object that wraps (.data):\n{self.data}"

    # ---- config --------------------------------------------------------------------------------------------
    def set_config(self, *, overwrite: bool = False, **kwargs):
        self.set_config_from_dict(kwargs, overwrite=overwrite)

    def set_config_from_dict(self, config_dict: Dict[str, Any], *, overwrite: bool = False):
        known = set(self._FEATURE_CONFIGS + self._LABEL_CONFIGS)
        if unknown := set(config_dict).difference(known):
            raise KeyError(f"Unknown config option(s): {unknown}, available options are: {known}")
        for group in (self._FEATURE_CONFIGS, self._LABEL_CONFIGS):
            vals = [v for k, v in config_dict.items() if k in group and v is not None]
            if len(set(map(type, vals))) > 1:
                raise TypeError(f"Found mixed types: {set(map(type, vals))}. Input configs must be either all str or all lists.")
            if vals and not isinstance(vals[0], str) and len(set(map(len, vals))) > 1:
                raise ValueError("Found mixed sizes lists. Input configs must be of same length.")
        for key, val in config_dict.items():
            if key not in self.config or self.config[key] == val:
                self.config[key] = val
            elif overwrite:
                self.config[key] = val
            else:
                raise KeyError(f"Config {key!r} exit with value {self.config[key]!r} but trying to set to a different "
                               f"value {val!r}. If you want to overwrite the config, please specify `overwrite=True`")

    # ---- splits --------------------------------------------------------------------------------------------
    def get_split_idx(self, split_name: str, error_on_miss: bool = False):
        if split_name is None:
            return list(range(self.shape[0]))
        if split_name in self._split_idx_dict:
            return self._split_idx_dict[split_name]
        if error_on_miss:
            raise KeyError(f"Unknown split {split_name!r}. Please set the split inddices via set_split_idx first.")
        return None

    def set_split_idx(self, split_name: str, split_idx: Sequence[int]):
        self._split_idx_dict[split_name] = split_idx

    def get_split_mask(self, split_name: str, return_type: str = "numpy"):
        """Boolean mask over the cells of one split (dance/data/base.py:341-360)."""
        split_idx = self.get_split_idx(split_name, error_on_miss=True)
        if return_type == "numpy":
            mask = np.zeros(self.shape[0], dtype=bool)
        elif return_type == "torch":
            mask = torch.zeros(self.shape[0], dtype=torch.bool)
        else:
            raise ValueError(f"Unsupported return_type {return_type!r}. Available options are 'numpy' and 'torch'.")
        mask[split_idx] = True
        return mask

    def get_split_data(self, split_name: str):
        return self._data[self.get_split_idx(split_name, error_on_miss=True)]

    def append(self, data, *, mode: Optional[str] = "merge", rename_dict: Optional[Dict[str, str]] = None,
               new_split_name: Optional[str] = None, label_batch: bool = False, **concat_kwargs):
        """Stack another data object under this one (dance/data/base.py:477-561).  ``mode`` says what happens to the new
        cells' splits: ``"merge"`` into the splits of the same name, ``"rename"`` through ``rename_dict``, ``"new_split"`` = all
        of them under ``new_split_name``, ``None`` = in no split.  ``label_batch`` numbers the appended blocks in
        ``obs["batch"]``."""
        import pandas as pd
        offset = self.shape[0]
        new_splits = {k: sorted(int(i) + offset for i in v) for k, v in data._split_idx_dict.items()}
        if mode == "merge":
            for name, idx in self._split_idx_dict.items():
                new_splits[name] = list(idx) + new_splits[name] if name in new_splits else idx
        elif mode == "rename":
            if rename_dict is None:
                raise ValueError("Mode 'rename' is selected but 'rename_dict' is not specified.")
            if common := set(self._split_idx_dict) & set(rename_dict.values()):
                raise ValueError(f"'rename_dict' cannot caontain split keys present in current data: {common}")
            if missed := [i for i in data._split_idx_dict if i not in rename_dict]:
                raise KeyError(f"Missing rename mapping for keys: {missed}")
            new_splits = {rename_dict[k]: v for k, v in new_splits.items()}
            new_splits.update(self._split_idx_dict)
        elif mode == "new_split":
            if new_split_name is None:
                raise ValueError("Mode 'new_split' is selected but 'new_split_name' is not specified.")
            if not isinstance(new_split_name, str):
                raise TypeError(f"'new_split_name' must be a string, got {type(new_split_name)}: {new_split_name}.")
            if new_split_name in self._split_idx_dict:
                raise ValueError(f"{new_split_name!r} is being used in the current splits. Please pick another name.")
            new_splits = {new_split_name: list(range(offset, offset + data.shape[0]))}
            new_splits.update(self._split_idx_dict)
        elif mode is None:
            new_splits = self._split_idx_dict
        else:
            raise ValueError(f"Unknown mode {mode!r}. Available options are: 'merge', 'rename', 'new_split'")
        new_uns = {k: v for k, v in data.data.uns.items() if not (isinstance(k, str) and k.endswith(".hip"))}
        new_uns.update({k: v for k, v in self._data.uns.items() if not (isinstance(k, str) and k.endswith(".hip"))})
        if label_batch:
            old = self._data.obs["batch"].tolist() if "batch" in self._data.obs.columns else [0] * self.shape[0]
            batch = list(map(int, old)) + [int(max(old)) + 1] * data.shape[0]
        if isinstance(self._data, AnnDataLite):
            self._data = concat((self._data, data.data), **concat_kwargs)
        else:  # a real AnnData was wrapped
            import anndata
            self._data = anndata.concat((self._data, data.data), **concat_kwargs)
        self._data.uns.update(new_uns)
        self._split_idx_dict = new_splits
        if label_batch:
            self._data.obs["batch"] = pd.Series(batch, dtype="category", index=self._data.obs.index)
        return self

    def pop(self, *, split_name: str):
        """Drop the cells of one split; the other splits are renumbered, emptied ones disappear (dance/data/base.py:563-577)."""
        gone = set(self.get_split_idx(split_name, error_on_miss=True))
        keep = [i for i in range(self.shape[0]) if i not in gone]
        new_pos = {j: i for i, j in enumerate(keep)}
        splits = {}
        for name, idx in self._split_idx_dict.items():
            if moved := sorted(new_pos[i] for i in idx if i in new_pos):
                splits[name] = moved
        self._data = self._data[keep]
        self._split_idx_dict = splits

    @property
    def train_idx(self):
        return self.get_split_idx("train", error_on_miss=False)

    @property
    def val_idx(self):
        return self.get_split_idx("val", error_on_miss=False)

    @property
    def test_idx(self):
        return self.get_split_idx("test", error_on_miss=False)

    # ---- features ------------------------------------------------------------------------------------------
    def _get_feature(self, channel, channel_type, mod):
        if mod is not None:
            raise AttributeError("`mod` needs a MuData object, which this container does not model")
        data = self._data
        if channel_type == "X":
            return data.X
        if channel_type == "raw_X":
            return data.raw.X
        if channel_type in ("obs", "var"):
            return getattr(data, channel_type)[channel]
        channel_type = channel_type or "obsm"
        if channel_type not in self._DATA_CHANNELS:
            raise ValueError(f"Unknown channel type {channel_type!r}. Available options are {self._DATA_CHANNELS}")
        if channel is None:
            warnings.warn("The `None` option for channel is deprecated; use channel_type='X'", DeprecationWarning, stacklevel=3)
            return data.X
        return getattr(data, channel_type)[channel]

    def get_feature(self, *, split_name: Optional[str] = None, return_type: str = "numpy",
                    channel: Optional[str] = None, channel_type: Optional[str] = "obsm", mod: Optional[str] = None):
        feature = self._get_feature(channel, channel_type, mod)
        channel_type = channel_type or "obsm"
        if isinstance(return_type, (torch.device, str)) and str(return_type).split(":")[0] in ("cuda", "cpu", "device"):
            # return_type = a torch device (or "device" = cuda): the fp32 matrix on that device, without a host round trip when
            # the slot already holds a DeviceArray; split rows are selected on the device
            dev = "cuda" if return_type == "device" else return_type
            t = to_device_matrix(feature, dev)
            if split_name is not None and channel_type in ["X", "raw_X", "obs", "obsm", "obsp", "layers"]:
                idx = torch.as_tensor([i for i in self.get_split_idx(split_name, error_on_miss=True) if i < t.shape[0]], device=t.device)
                t = t[idx][:, idx] if channel_type == "obsp" else t[idx]
            return t
        if isinstance(feature, DeviceArray) and return_type != "default":
            feature = feature.numpy()  # host consumers get a plain ndarray (materialised once)
        elif _is_lazy_graph(feature) and return_type != "default":
            feature = feature.materialize()  # graph.LazyScipyCSR -> the scipy matrix the reference keeps in obsp
        if return_type == "default":
            if split_name is not None:
                raise ValueError(f"split_name is not supported when return_type is 'default', got {split_name=!r}")
            return feature
        if return_type == "sparse":
            if isinstance(feature, np.ndarray):
                feature = sp.csr_matrix(feature)
            elif not sp.issparse(feature):
                raise ValueError(f"Feature is not sparse, got {type(feature)}")
        elif hasattr(feature, "toarray"):
            feature = feature.toarray()
        elif hasattr(feature, "to_numpy"):
            feature = feature.to_numpy()
        if split_name is not None:
            if channel_type in ["X", "raw_X", "obs", "obsm", "obsp", "layers"]:
                idx = [i for i in self.get_split_idx(split_name, error_on_miss=True) if i < feature.shape[0]]
                feature = feature[idx][:, idx] if channel_type == "obsp" else feature[idx]
        if return_type == "torch":
            feature = torch.from_numpy(feature)
        elif return_type not in ["numpy", "sparse"]:
            raise ValueError(f"Unknown return_type {return_type!r}")
        return feature

    def _get(self, config_keys, *, split_name=None, return_type="numpy", **kwargs):
        info = list(map(self.config.get, config_keys))
        if all(i is None for i in info):
            mods = channels = channel_types = [None]
        else:
            mods, channels, channel_types = map(_ensure_iter, info)
            n = max(len(mods), len(channels), len(channel_types))
            mods, channels, channel_types = (list(v) * n if len(v) == 1 and v[0] is None else v for v in (mods, channels, channel_types))
        out = []
        for m, c, t in zip(mods, channels, channel_types):
            try:
                out.append(self.get_feature(split_name=split_name, return_type=return_type, mod=m, channel=c, channel_type=t, **kwargs))
            except Exception as e:  # base.py:830-839: one error type for "this configured feature cannot be had"
                settings = dict(split_name=split_name, return_type=return_type, mod=m, channel=c, channel_type=t, kwargs=kwargs)
                raise RuntimeError(f"Failed to get features for the following settings:\n{pformat(settings)}") from e
        return out[0] if len(out) == 1 else out

    def get_x(self, split_name=None, return_type="numpy", **kwargs):
        return self._get(self._FEATURE_CONFIGS, split_name=split_name, return_type=return_type, **kwargs)

    def get_y(self, split_name=None, return_type="numpy", **kwargs):
        return self._get(self._LABEL_CONFIGS, split_name=split_name, return_type=return_type, **kwargs)

    def get_data(self, split_name=None, return_type="numpy", x_kwargs=dict(), y_kwargs=dict()):
        return self.get_x(split_name, return_type, **x_kwargs), self.get_y(split_name, return_type, **y_kwargs)

    def get_train_data(self, return_type="numpy", x_kwargs=dict(), y_kwargs=dict()):
        return self.get_data("train", return_type, x_kwargs, y_kwargs)

    def get_val_data(self, return_type="numpy", x_kwargs=dict(), y_kwargs=dict()):
        return self.get_data("val", return_type, x_kwargs, y_kwargs)

    def get_test_data(self, return_type="numpy", x_kwargs=dict(), y_kwargs=dict()):
        return self.get_data("test", return_type, x_kwargs, y_kwargs)
