"""Scope registry for drop-in discovery of transforms (mirrors the contract of dance/registry.py:190-233):
``@register_preprocessor("graph", "cell")`` files a class under ``preprocessor.graph.cell.<ClassName>`` so that a
pipeline ``Action(type="graph.cell", target="CellFeatureGraph")`` (dance/pipeline.py:105-125) resolves to it."""
from functools import partial
from typing import Any, Dict, Iterator, Optional


class DotDict(dict):
    """Nested dict whose levels can be reached as attributes or with one dotted key (dance/registry.py:9-93):
    ``d.a.b == d.get("a.b") == d["a"]["b"]``."""

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __init__(self, dictionary: Optional[Dict[str, Any]] = None):
        super().__init__()
        for key, val in (dictionary or {}).items():
            if "." in key:
                raise ValueError(f"key for DotDict cannot contain '.': {key!r}")
            self[key] = DotDict(val) if hasattr(val, "keys") else val

    def get(self, key: str, default: Any = None, missed_ok: bool = True, create_on_miss: bool = False):
        """The entry under the dotted ``key`` ("" = this level).  A missing entry gives ``default``, raises (``missed_ok=False``) or is
        created as an empty level (``create_on_miss``)."""
        if create_on_miss and not missed_ok:
            raise ValueError("missed_ok must be enabled when create_on_miss is enabled.")
        if key == "":
            return self
        parts = key.split(".")
        node = self
        for part in parts:
            if not isinstance(node, dict) or part not in node:
                if create_on_miss:
                    made = DotDict()
                    self.set(key, made)
                    return made
                if missed_ok:
                    return default
                raise KeyError(f"Failed to decode keys {parts!r}")
            node = node[part]
        return node

    def set(self, key: str, val: Any, exist_ok: bool = True):
        if not exist_ok and self.get(key) is not None:
            raise KeyError(f"Key exists: {key}")
        *scope, leaf = key.split(".")
        node = self
        for i, part in enumerate(scope):
            node = node.setdefault(part, DotDict())
            if not isinstance(node, DotDict):
                raise KeyError(f"Level {i} ({part!r}) is already set as a non-leaf node: {node}.")
        node[leaf] = val


class Registry(DotDict):
    """The scope tree the transforms / metrics register into (dance/registry.py:96-160)."""

    def is_leaf_node(self, key: str) -> bool:
        return not isinstance(self.get(key), DotDict)

    def children(self, key: str = "", leaf_node: bool = True, non_leaf_node: bool = True, return_val: bool = False) -> Iterator[Any]:
        """Dotted keys (or ``(key, value)`` pairs) of everything below the level ``key``, depth first in insertion order; leaves
        and / or inner levels."""
        if not non_leaf_node and not leaf_node:
            raise ValueError("Either one, or both, of leaf_node and non_leaf_node must be True")
        try:
            top = self.get(key, missed_ok=False)
        except KeyError:
            raise KeyError(f"{key!r} node does not exist yet.")
        if not isinstance(top, DotDict):
            raise KeyError(f"{key} is a leaf node. children only take non-leaf nodes.")
        stack = [(f"{key}.{name}".lstrip("."), val) for name, val in reversed(list(top.items()))]
        while stack:
            path, val = stack.pop()
            inner = isinstance(val, DotDict)
            if non_leaf_node if inner else leaf_node:
                yield (path, val) if return_val else path
            if inner:
                stack.extend((f"{path}.{name}", v) for name, v in reversed(list(val.items())))


REGISTRY = Registry()


def register(*scope: str, name: Optional[str] = None, overwrite: bool = False, _registry: Registry = REGISTRY):

    def wrap(obj):
        obj_name = name or obj.__name__
        key = ".".join((*scope, obj_name))
        try:
            _registry.set(key, obj, exist_ok=overwrite)
        except KeyError as e:
            if _registry.get(key) != obj:
                raise KeyError(f"{obj_name!r} already registered under {scope}") from e
        return obj

    return wrap


REGISTRY_PREFIX = "_registry_"


def resolve_from_registry(name: str, scope: str, registry: Registry = REGISTRY):
    """``registry[scope][name]``; a leading ``_registry_.`` in the scope (the pipeline configs write it) is skipped."""
    scope = scope.replace(REGISTRY_PREFIX, "", 1).lstrip(".")
    return registry.get(".".join((scope, name)), missed_ok=False)


register_preprocessor = partial(register, "preprocessor")
register_metric_func = partial(register, "function", "metric")
