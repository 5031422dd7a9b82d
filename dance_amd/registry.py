"""Scope registry for drop-in discovery of transforms (mirrors the contract of dance/registry.py:190-233):
``@register_preprocessor("graph", "cell")`` files a class under ``preprocessor.graph.cell.<ClassName>`` so that a
pipeline ``Action(type="graph.cell", target="CellFeatureGraph")`` (dance/pipeline.py:105-125) resolves to it."""
from functools import partial
from typing import Any, Optional


class Registry(dict):
    """Nested dict addressed with dotted keys."""

    def get(self, key: str, default: Any = None, missed_ok: bool = True):
        node = self
        if key == "":
            return node
        for part in key.split("."):
            if not isinstance(node, dict) or part not in node:
                if missed_ok:
                    return default
                raise KeyError(f"Failed to decode keys {key.split('.')!r}")
            node = node[part]
        return node

    def set(self, key: str, val: Any, exist_ok: bool = True):
        if not exist_ok and self.get(key) is not None:
            raise KeyError(f"Key exists: {key}")
        *scope, leaf = key.split(".")
        node = self
        for i, part in enumerate(scope):
            node = node.setdefault(part, Registry())
            if not isinstance(node, dict):
                raise KeyError(f"Level {i} ({part!r}) is already set as a non-leaf node: {node}.")
        node[leaf] = val

    def is_leaf_node(self, key: str) -> bool:
        return not isinstance(self.get(key), dict)


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


def resolve_from_registry(name: str, scope: str, registry: Registry = REGISTRY):
    return registry.get(".".join((scope, name)), missed_ok=False)


register_preprocessor = partial(register, "preprocessor")
register_metric_func = partial(register, "function", "metric")
