"""Method protocol (mirrors dance/modules/base.py:17-199): preprocessing_pipeline / fit / predict / predict_proba /
score / fit_predict / fit_score, default metrics ``acc`` (classification, dance/utils/metrics.py:33-58), ``ari``
(clustering, :61-70) and ``mse`` (regression, :71-80)."""
from abc import ABC, abstractmethod
from typing import Any, Mapping, Optional, Tuple, Union

import numpy as np
import torch


def _acc(true, pred) -> float:
    """utils/metrics.py:33-58: ``true`` is a (possibly multi-label) one-hot matrix or a label vector."""
    true = true.cpu().numpy() if isinstance(true, torch.Tensor) else np.asarray(true)
    pred = pred.cpu().numpy() if isinstance(pred, torch.Tensor) else np.asarray(pred)
    if true.ndim == 2:
        return float(true[np.arange(pred.shape[0]), pred.ravel().astype(int)].sum() / pred.shape[0])
    return float((true == pred).mean())


def _ari(true, pred) -> float:
    from sklearn.metrics import adjusted_rand_score
    return float(adjusted_rand_score(np.asarray(true).ravel(), np.asarray(pred).ravel()))


def _mse(true, pred) -> float:
    """utils/metrics.py:71-80 (``sklearn.metrics.mean_squared_error``)."""
    true = true.cpu().numpy() if isinstance(true, torch.Tensor) else np.asarray(true)
    pred = pred.cpu().numpy() if isinstance(pred, torch.Tensor) else np.asarray(pred)
    return float(np.mean((true.astype(np.float64) - pred.astype(np.float64))**2))


_METRICS = {"acc": _acc, "ari": _ari, "mse": _mse}


def resolve_score_func(score_func):
    if isinstance(score_func, str):
        return _METRICS[score_func]
    if callable(score_func):
        return score_func
    raise TypeError(f"Unknown type {type(score_func)}, must be str or callable")


class BaseMethod(ABC):

    _DEFAULT_METRIC: Optional[str] = None
    _DISPLAY_ATTRS: Tuple[str] = ()

    @property
    def name(self):
        return type(self).__name__

    def __repr__(self) -> str:
        return f"{self.name}({', '.join(f'{i}={getattr(self, i)!r}' for i in self._DISPLAY_ATTRS)})"

    def preprocess(self, data, /, **kwargs):
        self.preprocessing_pipeline(**kwargs)(data)

    @staticmethod
    @abstractmethod
    def preprocessing_pipeline(**kwargs):
        ...

    @abstractmethod
    def fit(self, x, y, **kwargs):
        ...

    def predict_proba(self, x):
        raise NotImplementedError

    @abstractmethod
    def predict(self, x):
        ...

    @property
    def default_score_func(self) -> Mapping[Any, float]:
        return resolve_score_func(self._DEFAULT_METRIC)

    def score(self, x, y, *, score_func=None, return_pred: bool = False) -> Union[float, Tuple[float, Any]]:
        y_pred = self.predict(x)
        score = resolve_score_func(score_func or self._DEFAULT_METRIC)(y, y_pred)
        return (score, y_pred) if return_pred else score

    def fit_predict(self, x, y=None, **fit_kwargs):
        self.fit(x, y, **fit_kwargs)
        return self.predict(x)

    def fit_score(self, x, y, *, score_func=None, return_pred: bool = False, **fit_kwargs):
        self.fit(x, **fit_kwargs)
        return self.score(x, y, score_func=score_func, return_pred=return_pred)


class BasePretrain(ABC):
    """dance/modules/base.py:72-117: pre-train once (or load ``pretrain_path``), remember it, save the result."""

    @property
    def is_pretrained(self) -> bool:
        return getattr(self, "_is_pretrained", False)

    def _pretrain(self, *args, force_pretrain: bool = False, **kwargs):
        import os
        pt_path = getattr(self, "pretrain_path", None)
        if not force_pretrain:
            if self.is_pretrained:
                return
            if pt_path is not None and os.path.isfile(pt_path):
                self.load_pretrained(pt_path)
                self._is_pretrained = True
                return
        self.pretrain(*args, **kwargs)
        self._is_pretrained = True
        if pt_path is not None:
            self.save_pretrained(pt_path)

    def pretrain(self, *args, **kwargs):
        ...

    def save_pretrained(self, path, **kwargs):
        ...

    def load_pretrained(self, path, **kwargs):
        ...


class TorchNNPretrain(BasePretrain, ABC):
    """dance/modules/base.py:120-153: lock / unlock sub-modules (``requires_grad``) around pre-training."""

    def _fix_unfix_modules(self, *module_names, unfix: bool = False, single: bool = True):
        from operator import attrgetter
        modules = attrgetter(*module_names)(self)
        for module in ([modules] if single else modules):
            for p in module.parameters():
                p.requires_grad = unfix

    def fix_module(self, *names):
        self._fix_unfix_modules(*names, unfix=False, single=True)

    def fix_modules(self, *names):
        self._fix_unfix_modules(*names, unfix=False, single=False)

    def unfix_module(self, *names):
        self._fix_unfix_modules(*names, unfix=True, single=True)

    def unfix_modules(self, *names):
        self._fix_unfix_modules(*names, unfix=True, single=False)

    def pretrain_context(self, *module_names):
        from contextlib import contextmanager

        @contextmanager
        def ctx():
            single = len(module_names) == 1
            self._fix_unfix_modules(*module_names, unfix=True, single=single)
            try:
                yield
            finally:
                self._fix_unfix_modules(*module_names, unfix=False, single=single)
        return ctx()

    def save_pretrained(self, path):
        torch.save(self.state_dict(), path)

    def load_pretrained(self, path):
        self.load_state_dict(torch.load(path, map_location=getattr(self, "device", None)))


class BaseClassificationMethod(BaseMethod):
    _DEFAULT_METRIC = "acc"


class BaseRegressionMethod(BaseMethod):

    _DEFAULT_METRIC = "mse"


class BaseClusteringMethod(BaseMethod):
    _DEFAULT_METRIC = "ari"

    def score(self, x, y, *, score_func=None, return_pred: bool = False, valid_idx=None, test_idx=None):
        y_pred = self.predict(x)
        func = resolve_score_func(score_func or self._DEFAULT_METRIC)
        if valid_idx is None:
            score = func(y, y_pred)
        else:  # dance/modules/base.py:177-186
            score = {"valid_score": func([y[i] for i in valid_idx], [y_pred[i] for i in valid_idx]),
                     "test_score": func([y[i] for i in test_idx], [y_pred[i] for i in test_idx])}
        return (score, y_pred) if return_pred else score
