"""Method protocol (mirrors dance/modules/base.py:17-199): preprocessing_pipeline / fit / predict / predict_proba /
score / fit_predict / fit_score, default metrics ``acc`` (classification, dance/utils/metrics.py:33-58) and ``ari``
(clustering, :61-70)."""
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


_METRICS = {"acc": _acc, "ari": _ari}


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


class BaseClassificationMethod(BaseMethod):
    _DEFAULT_METRIC = "acc"


class BaseClusteringMethod(BaseMethod):
    _DEFAULT_METRIC = "ari"

    def score(self, x, y, *, score_func=None, return_pred: bool = False, valid_idx=None, test_idx=None):
        y_pred = self.predict(x)
        func = resolve_score_func(score_func or self._DEFAULT_METRIC)
        if valid_idx is None:
            score = func(y, y_pred)
        else:
            score = func(np.asarray(y)[valid_idx], np.asarray(y_pred)[valid_idx])
        return (score, y_pred) if return_pred else score
