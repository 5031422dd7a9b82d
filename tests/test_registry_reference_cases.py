"""CPU: the scope registry the operators register into, against the known-answer cases the reference holds for its own
(tests/test_registry.py:8-103: dotted access, creation on miss, leaf / inner children, scoped registration), plus what this repo's
transforms actually registered."""
from functools import partial

import pytest

from dance_amd.registry import REGISTRY, DotDict, Registry, register, resolve_from_registry


def test_dotdict():
    DotDict()
    dd = DotDict({"a": 1, "b": 2, "c": 3})
    assert dd["a"] == dd.a == 1 and dd["b"] == dd.b == 2 and dd["c"] == dd.c == 3
    dd = DotDict({"a": {"b": {"c": 1}}})
    assert dd.a.b.c == dd["a"]["b"]["c"] == 1
    assert dd.a.b.c == dd.a.b["c"] == dd["a"].b.c == dd.a["b"].c
    assert dd.get("a.b.c") == dd.get("a.b")["c"] == dd.get("a.b").c == 1
    assert dd.get("a.b.d") is None
    with pytest.raises(KeyError):
        dd.get("a.b.d", missed_ok=False)
    node = dd.get("x.y.z", create_on_miss=True)
    assert dict(dd.x.y.z) == dict()
    dd.x.y.z["test"] = 2  # the returned node is the stored one
    assert dict(node) == dict(dd.x.y.z) == dict(test=2)
    with pytest.raises(ValueError):
        dd.get("d.e", missed_ok=False, create_on_miss=True)
    dd = DotDict({"a": {"b": {"c": 1}}})
    dd.set("a.b.d", 2)
    assert dd.get("a.b.d", missed_ok=False) == 2
    dd.set("a.b.d", 3, exist_ok=True)
    assert dd.get("a.b.d", missed_ok=False) == 3
    with pytest.raises(KeyError):
        dd.set("a.b.d", 3, exist_ok=False)
    with pytest.raises(KeyError):
        dd.set("a.b.c.d", 4)
    with pytest.raises(ValueError):
        DotDict({"a.b": 1})


def test_registry_children():
    Registry()
    r = Registry({"a": 1, "b": {"c": 2}})
    assert r.is_leaf_node("a") and not r.is_leaf_node("b") and r.is_leaf_node("b.c")
    assert sorted(r.children(leaf_node=True, non_leaf_node=True)) == ["a", "b", "b.c"]
    assert sorted(r.children("b", leaf_node=True, non_leaf_node=True)) == ["b.c"]
    assert sorted(r.children(leaf_node=False, non_leaf_node=True)) == ["b"]
    assert sorted(r.children(leaf_node=True, non_leaf_node=False)) == ["a", "b.c"]
    with pytest.raises(KeyError):
        list(r.children("a"))
    with pytest.raises(KeyError):
        list(r.children("zzz"))
    with pytest.raises(ValueError):
        list(r.children(leaf_node=False, non_leaf_node=False))
    assert list(r.children(leaf_node=True, non_leaf_node=False, return_val=True)) == [("a", 1), ("b.c", 2)]
    deep = Registry({"p": {"q": {"r": 1, "s": 2}, "t": 3}, "u": 4})
    assert list(deep.children()) == ["p", "p.q", "p.q.r", "p.q.s", "p.t", "u"]          # depth first, insertion order


def test_register():
    r1, r2 = Registry(), Registry()
    register("a", name="test", _registry=r1)(1)
    assert dict(r1) == {"a": {"test": 1}} and dict(r2) == {}
    register("a.b", name="test", _registry=r1)(2)
    assert dict(r1) == {"a": {"test": 1, "b": {"test": 2}}}
    register("a", "c", "d", name="test", _registry=r1)(3)
    assert dict(r1) == {"a": {"test": 1, "b": {"test": 2}, "c": {"d": {"test": 3}}}} and dict(r2) == {}
    partial(register, "b")("c", "d", name="test", _registry=r2)(4)
    assert dict(r2) == {"b": {"c": {"d": {"test": 4}}}}
    assert resolve_from_registry("test", "_registry_.a.b", r1) == resolve_from_registry("test", "a.b", r1) == 2


def test_what_the_transforms_registered():
    import dance_amd.transforms  # noqa: F401  (registration happens at import)
    import dance_amd.transforms.graph  # noqa: F401
    leaves = set(REGISTRY.children("preprocessor", non_leaf_node=False))
    for key in ("preprocessor.graph.cell.NeighborGraph", "preprocessor.graph.feature.FeatureFeatureGraph", "preprocessor.filter.gene.FilterGenesTopK",
                "preprocessor.filter.gene.HighlyVariableGenesRawCount", "preprocessor.filter.cell.FilterCellsScanpyOrder",
                "preprocessor.split.entry.CellwiseMaskData", "preprocessor.misc.UpdateRaw"):
        assert key in leaves, key
    assert REGISTRY.preprocessor.graph.spatial.SpaGCNGraph.__name__ == "SpaGCNGraph"
