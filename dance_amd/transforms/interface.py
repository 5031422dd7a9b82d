"""``AnnDataTransform`` (dance/transforms/interface.py:8-62): run any function that edits an AnnData-like object in place — given as a
callable or as its dotted import path — on ``data.data``.  The reference wraps the ``scanpy.pp`` functions with it; the device-native
steps of this package (``NormalizeTotal``, ``Log1P``, the HVG transforms ...) replace those, and this interface is what lets a user's
own host function sit in the same ``Compose``."""
import importlib
from typing import Callable, Union

from ..registry import register_preprocessor
from .base import BaseTransform


@register_preprocessor("interface")
class AnnDataTransform(BaseTransform):

    _DISPLAY_ATTRS = ("func", "func_kwargs")

    def __init__(self, func: Union[Callable, str], **kwargs):
        super().__init__()
        self.func = func
        self.func_kwargs = kwargs

    @property
    def func(self) -> Callable:
        return self._func

    @func.setter
    def func(self, func: Union[Callable, str]):
        if isinstance(func, str):
            scope, _, attr = func.rpartition(".")
            func = getattr(importlib.import_module(scope), attr)
        if not callable(func):
            raise TypeError(f"Interfaced function must be callable, got {type(func)}: {func!r}")
        self._func = func

    def __repr__(self):
        return f"{self.name}(func={self.func.__module__}.{self.func.__name__}, func_kwargs={self.func_kwargs})"

    def __call__(self, data):
        self.func(data.data, **self.func_kwargs)
