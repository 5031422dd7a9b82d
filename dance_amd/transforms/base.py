"""Transform protocol (mirrors dance/transforms/base.py:12-49): ``out`` defaults to the class name, ``__repr__`` is
built from ``_DISPLAY_ATTRS`` (pinned by the reference's tests/transforms/test_basics.py:5-30), ``hexdigest`` is
md5(repr) (dataset cache key, dance/datasets/base.py:129-133), ``__call__(data)`` mutates ``data`` in place."""
import hashlib
import logging
from abc import ABC, abstractmethod
from typing import Optional, Tuple

logger = logging.getLogger("dance")


class BaseTransform(ABC):

    _DISPLAY_ATTRS: Tuple[str] = ()

    def __init__(self, out: Optional[str] = None, log_level="WARNING"):
        self.out = out or self.name
        self.logger = logger.getChild(self.name)
        self.logger.setLevel(log_level)
        self.log_level = log_level

    @property
    def name(self) -> str:
        return self.__class__.__name__

    def hexdigest(self) -> str:
        return hashlib.md5(repr(self).encode()).hexdigest()

    def __repr__(self) -> str:
        attrs = ", ".join(f"{i}={getattr(self, i)!r}" for i in self._DISPLAY_ATTRS)
        return f"{self.name}({attrs})"

    @abstractmethod
    def __call__(self, data):
        raise NotImplementedError
