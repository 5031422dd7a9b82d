"""CPU: the known-answer cases the reference holds for its data object (tests/data/test_data.py:12-194 — splits, feature
retrieval per channel type, append modes, batch labels), run on dance_amd.data.Data over the AnnData stand-in; plus pop /
split masks / concat by variable name (dance/data/base.py:341-372, 563-577)."""
import numpy as np
import pandas as pd
import pytest

from dance_amd.data import AnnDataLite as AnnData
from dance_amd.data import Data

X = np.array([[0, 1], [1, 2], [2, 3]], dtype=np.float32)
Y = np.array([[0], [1], [2]], dtype=np.float32)


def mk():
    return AnnData(X.copy())


def test_basic():
    d = Data(mk())
    assert d.num_cells == 3 and d.num_features == 2
    assert d.cells == ["0","1","2"]
    assert d.train_idx is d.val_idx is d.test_idx is None
    d = Data(mk(), train_size="all")
    assert d.train_idx == [0,1,2] and d.val_idx is None and d.test_idx is None
    d = Data(mk(), train_size=2)
    assert d.train_idx == [0,1] and d.test_idx == [2]
    d = Data(mk(), train_size=-1, test_size=1)
    assert d.train_idx == [0,1] and d.test_idx == [2]
    d = Data(mk(), train_size=1, val_size=1)
    assert (d.train_idx, d.val_idx, d.test_idx) == ([0],[1],[2])
    with pytest.raises(TypeError):
        Data(mk(), train_size="1")
    with pytest.raises(ValueError):
        Data(mk(), train_size=-1)
    with pytest.raises(ValueError):
        Data(mk(), train_size=5)
    with pytest.raises(ValueError):
        Data(mk(), train_size=2, test_size=2)
    d = Data(mk(), split_index_range_dict={"train": (0, 1), "ref": (0, 2), "inf": (2, 3)})
    assert d.train_idx == [0] and d.get_split_idx("ref") == [0,1] and d.get_split_idx("inf") == [2]
    for bad in ((0,1,2), [0,1], ("0","1")):
        with pytest.raises(TypeError):
            Data(mk(), split_index_range_dict={"train": bad})
    d = Data(mk(), full_split_name="inference")
    assert d.train_idx is None and d.get_split_idx("inference") == [0,1,2]


def test_get():
    def mk2():
        a = AnnData(X.copy(), obs=pd.DataFrame(X, columns=["a","b"]), var=pd.DataFrame(X.T, columns=["x","y","z"]))
        a.obsm["feature1"] = X+10
        a.obsm["feature2"] = X+20
        a.layers["layer_feature"] = X+30
        a.obsm["obsm_feature"] = X
        a.obsp["obsp_feature"] = X@X.T
        a.varm["varm_feature"] = X.T
        a.varp["varp_feature"] = X.T@X
        a.obsm["label"] = Y
        return a
    d = Data(mk2(), train_size=2)
    d.set_config(label_channel="label")
    x,y = d.get_train_data()
    assert x.tolist()==[[0,1],[1,2]] and y.tolist()==[[0],[1]]
    x,y = d.get_test_data()
    assert x.tolist()==[[2,3]] and y.tolist()==[[2]]
    pytest.raises(RuntimeError, d.get_val_data)
    d = Data(mk2(), train_size=2)
    d.set_config(feature_channel=[None,"feature1","feature2"], label_channel="label")
    (x1,x2,x3),y = d.get_train_data()
    assert x2.tolist()==[[10,11],[11,12]] and x3.tolist()==[[20,21],[21,22]]
    d = Data(mk2(), train_size=2)
    d.set_config(feature_channel=["obsm_feature","obsp_feature","varm_feature","varp_feature","layer_feature"],
                 feature_channel_type=["obsm","obsp","varm","varp","layers"], label_channel="label")
    (a,b,c,e,f),y = d.get_train_data()
    assert a.tolist()==[[0,1],[1,2]] and b.tolist()==[[1,2],[2,5]] and c.tolist()==[[0,1,2],[1,2,3]]
    assert e.tolist()==[[5,8],[8,14]] and f.tolist()==[[30,31],[31,32]]
    d = Data(mk2(), train_size=2)
    d.set_config(feature_channel=["a","z"], feature_channel_type=["obs","var"], label_channel="label")
    (x1,x2),_ = d.get_train_data()
    assert x1.tolist()==[0,1] and x2.tolist()==[2,3]


def test_append():
    d1 = Data(mk(), train_size=1)
    d2 = Data(mk(), train_size=2)
    s2 = {"train":[0,1],"test":[2]}
    d = d1.copy()
    d.append(d2, mode="merge")
    assert d._split_idx_dict == {"train":[0,3,4],"test":[1,2,5]} and d2._split_idx_dict == s2
    pytest.raises(ValueError, d1.copy().append, d2, mode="rename")
    pytest.raises(KeyError, d1.copy().append, d2, mode="rename", rename_dict={"train":"new"})
    pytest.raises(ValueError, d1.copy().append, d2, mode="rename", rename_dict={"train":"a","test":"test"})
    d = d1.copy()
    d.append(d2, mode="rename", rename_dict={"train":"new_train","test":"new_test"})
    assert d._split_idx_dict == {"train":[0],"new_train":[3,4],"test":[1,2],"new_test":[5]}
    pytest.raises(ValueError, d1.copy().append, d2, mode="new_split")
    pytest.raises(ValueError, d1.copy().append, d2, mode="new_split", new_split_name="test")
    d = d1.copy()
    d.append(d2, mode="new_split", new_split_name="ref")
    assert d._split_idx_dict == {"train":[0],"test":[1,2],"ref":[3,4,5]}
    d = d1.copy()
    d.append(d2, mode=None)
    assert d._split_idx_dict == {"train":[0],"test":[1,2]} and "batch" not in d.data.obs
    d = d1.copy()
    d.append(d2, mode=None, label_batch=True)
    assert d.data.obs["batch"].tolist()==[0,0,0,1,1,1]
    d.append(d2, mode=None, label_batch=True)
    assert d.data.obs["batch"].tolist()==[0,0,0,1,1,1,2,2,2]


def test_pop_masks_and_concat_by_name():
    d = Data(mk(), train_size=1, val_size=1)
    assert d.get_split_mask("val").tolist() == [False, True, False]
    assert d.get_split_mask("test", return_type="torch").tolist() == [False, False, True]
    with pytest.raises(KeyError):
        d.get_split_mask("nope")
    assert d.get_split_data("test").X.tolist() == [[2, 3]] and d[[0, 2]].obs.index.tolist() == ["0", "2"]
    d.pop(split_name="val")
    assert d._split_idx_dict == {"train": [0], "test": [1]} and d.data.X.tolist() == [[0, 1], [2, 3]] and d.cells == ["0", "2"]
    # variables are matched by name: the inner join keeps the shared ones in the first part's order
    a = AnnData(X.copy(), var=pd.DataFrame(index=["g1", "g2"]), obsm={"e": X, "only_a": X})
    b = AnnData(X[:, ::-1] * 10, var=pd.DataFrame(index=["g2", "g1"]), obsm={"e": X + 1})
    da = Data(a, train_size=2).append(Data(b, train_size=1), mode="merge")
    assert da.data.X.tolist() == X.tolist() + (X * 10).tolist() and list(da.data.obsm) == ["e"] and da.data.obsm["e"].shape == (6, 2)
    assert da.train_idx == [0, 1, 3] and da.test_idx == [2, 4, 5] and da.data.var.index.tolist() == ["g1", "g2"]
    assert "dance_config" in da.data.uns
    with pytest.raises(KeyError):
         # unknown config keys are refused (base.py:238-240)
        da.set_config(feature="e")
    with pytest.raises(TypeError):
         # str with list (base.py:33-41)
        da.set_config(feature_channel=["e"], feature_channel_type="obsm")


def test_filter_by_mask_returns_sorted_splits():
    """dance/data/base.py:780 stores sorted(new_indices) for every split after a cell filter, whatever order the split was
    given in (the remapped lists used to keep the original order)."""
    d = Data(AnnData(np.arange(12, dtype=np.float32).reshape(6, 2)))
    d._split_idx_dict = {"train": [4, 0, 2], "val": [5, 1], "test": [3]}
    d.filter_by_mask(np.array([True, False, True, False, True, True]))
    assert d._split_idx_dict == {"train": [0, 1, 2], "val": [3], "test": []}
