"""CPU: dance's own gene filters (FilterGenesPercentile / FilterGenesTopK / FilterGenesCommon, filter.py:319-383, 437-662) against
what the reference's ``__call__`` bodies did on the same inputs (tests/golden/gene_filters.npz, tests/golden/make_golden.py
``make_gene_filters``), the reference's own known-answer case (tests/transforms/test_filter.py:10-31) and the ordered scanpy
filters (filter.py:1048-1139, 1403-1473; tests/transforms/test_filter_cell_gene.py:30-72) against the numpy restatement applied
step by step.  The statistics are device reductions: ``check_gene_filters("cuda")`` is the GPU twin (tests/test_gpu_transforms.py)."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from oracle import normalize as on

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gene_filters.npz")


def check_gene_filters(device):
    from dance_amd.data import AnnDataLite, Data, DeviceArray, concat
    from dance_amd.transforms import (FilterCellsScanpyOrder, FilterGenesCommon, FilterGenesPercentile, FilterGenesScanpyOrder,
                                      FilterGenesTopK)
    g = np.load(GOLD)
    x, names = g["gf_x"], [str(v) for v in g["gf_names"]]

    def mk(slot=None, **kw):
        slot = DeviceArray(torch.from_numpy(x.copy()).to(device)) if slot is None else slot
        return Data(AnnDataLite(slot, var=pd.DataFrame(index=names), obs=pd.DataFrame(index=[f"c{i}" for i in range(len(x))]), **kw))

    def same(d, tag):
        want = [str(v) for v in g[f"gf_{tag}_selected"]]
        cols = [names.index(v) for v in want]
        assert list(d.data.var_names) == want, tag                      # the same genes, in the reference's (name-sorted) order
        assert isinstance(d.data.X, DeviceArray) and np.array_equal(d.data.X.tensor.cpu().numpy(), x[:, cols])
        assert np.allclose(d.data.uns["gene_summary"], g[f"gf_{tag}_summary"], rtol=2e-5, atol=1e-6), tag  # the reference sums in fp32
        v = d.data.var
        assert np.array_equal(np.asarray(v["n_counts"], dtype=np.float64), g[f"gf_{tag}_n_counts"][cols])
        assert np.array_equal(np.asarray(v["n_cells"]), g[f"gf_{tag}_n_cells"][cols])

    before = DeviceArray.host_copies
    for mode in ("sum", "var", "cv", "rv"):
        d = mk()
        FilterGenesPercentile(min_val=10, max_val=90, mode=mode, device=device)(d)
        same(d, f"pct_{mode}")
        d = mk()
        FilterGenesTopK(num_genes=12, mode=mode, add_n_counts=True, add_n_cells=True, device=device)(d)
        same(d, f"top_{mode}")
    for tag, t in (("pct_default", FilterGenesPercentile(device=device)),
                   ("top_all", FilterGenesTopK(99, mode="sum", add_n_counts=True, add_n_cells=True, device=device)),
                   ("bottom_cv", FilterGenesTopK(7, top=False, add_n_counts=True, add_n_cells=True, device=device))):
        d = mk()
        t(d)
        same(d, tag)
    assert DeviceArray.host_copies == before                              # the matrix never left the device
    d = mk()
    d.data.var["keep_me"] = np.isin(np.arange(len(names)), [3, 17])
    FilterGenesPercentile(min_val=10, max_val=90, whitelist_indicators="keep_me", device=device)(d)
    assert list(d.data.var_names) == [str(v) for v in g["gf_pct_whitelist_selected"]] and "g3" in d.data.var_names
    d = mk()
    t = FilterGenesTopK(5, mode="var", inplace=False, device=device)
    t(d)
    top5 = sorted(np.array(names)[np.argsort(g["gf_top_var_summary"])[-5:]])
    assert d.data.X.shape == x.shape and np.array_equal(np.asarray(d.data.obsm[t.out]), x[:, [names.index(v) for v in top5]])
    with pytest.raises(ValueError):
        FilterGenesTopK(mode="median")
    with pytest.raises(ValueError):
        FilterGenesPercentile(channel="counts")                            # a channel needs channel_type="layers"
    d = mk(layers={"counts": DeviceArray(torch.from_numpy(x.copy() * 2).to(device))})
    FilterGenesPercentile(min_val=10, max_val=90, channel="counts", channel_type="layers", device=device)(d)
    assert list(d.data.var_names) == [str(v) for v in g["gf_pct_sum_selected"]]
    assert np.allclose(d.data.uns["gene_summary"], 2 * g["gf_pct_sum_summary"])

    # ---- FilterGenesCommon: goldens (dense == sparse in the reference), by batch column and by split ------------------------------
    assert list(g["gf_common_batch_dense"]) == list(g["gf_common_batch_sparse"])
    assert list(g["gf_common_split_dense"]) == list(g["gf_common_split_sparse"])
    batch = np.r_[np.zeros(30, int), np.ones(20, int), np.full(10, 2)]
    d = mk()
    d.data.obs["batch"] = batch
    FilterGenesCommon(batch_key="batch", device=device)(d)
    assert list(d.data.var_names) == [str(v) for v in g["gf_common_batch_dense"]]
    d = Data(mk().data, split_index_range_dict={"train": (0, 30), "test": (30, 60)})
    FilterGenesCommon(split_keys=["train", "test"], device=device)(d)
    want = [str(v) for v in g["gf_common_split_dense"]]
    assert list(d.data.var_names) == want and np.array_equal(np.asarray(d.data.X), x[:, [names.index(v) for v in want]])
    with pytest.raises(ValueError):
        FilterGenesCommon()
    with pytest.raises(ValueError):
        FilterGenesCommon(batch_key="batch", split_keys=["train"])
    with pytest.raises(KeyError):
        FilterGenesCommon(split_keys=["nope"], device=device)(mk())
    # the reference's own test (tests/transforms/test_filter.py:10-31): two batches, only gene "y" is expressed in both
    for mode in ("batch", "split"):
        var = pd.DataFrame(index=["x", "y", "z"])
        a1 = AnnDataLite(np.array([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=np.float32), obs=pd.DataFrame({"batch": 0}, index=["a", "b", "c"]), var=var)
        a2 = AnnDataLite(np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float32), obs=pd.DataFrame({"batch": 1}, index=["d", "e"]), var=var)
        d = Data(concat((a1, a2)), train_size=3)
        t = FilterGenesCommon(batch_key="batch", device=device) if mode == "batch" else FilterGenesCommon(split_keys=["train", "test"], device=device)
        t(d)
        assert d.shape[1] == 1 and list(d.data.var_names) == ["y"]

    # ---- ordered scanpy filters: the reference's test matrix and orders, against the restated single filters applied in turn ------
    toy = (np.random.default_rng(123).random((50, 30)) * 10).astype(np.float32)
    kw = dict(min_counts=1, min_cells=1, max_counts=3000, max_cells=20)
    for order in (["min_counts", "min_cells", "max_counts", "max_cells"], ["max_counts", "min_cells", "min_counts"], ["min_cells", "min_counts"],
                  ["min_counts"], []):
        d = Data(AnnDataLite(DeviceArray(torch.from_numpy(toy.copy()).to(device))))
        FilterGenesScanpyOrder(order=order, device=device, **kw)(d)
        ans = toy
        for key in order:
            ans = ans[:, on.filter_genes(ans, **{key: kw[key]})[0]]
        assert d.data.X.shape == ans.shape and np.array_equal(np.asarray(d.data.X), ans), order
    kw = dict(min_counts=1, min_genes=1, max_counts=3000, max_genes=20)
    for order in (["min_counts", "min_genes", "max_counts", "max_genes"], ["max_counts", "min_genes", "min_counts"], ["min_genes", "min_counts"],
                  ["min_counts"], []):
        d = Data(AnnDataLite(DeviceArray(torch.from_numpy(toy.copy()).to(device))))
        FilterCellsScanpyOrder(order=order, device=device, **kw)(d)
        ans = toy
        for key in order:
            ans = ans[on.filter_cells(ans, **{key: kw[key]})[0]]
        assert d.data.X.shape == ans.shape and np.array_equal(np.asarray(d.data.X), ans), order
    assert FilterGenesScanpyOrder(min_cells=2).filter_genes_order == ["min_counts", "min_cells", "max_counts", "max_cells"]
    with pytest.raises(KeyError):
        FilterGenesScanpyOrder(order=["min_genes"], min_counts=1)


def test_gene_filters_on_cpu_tensors():
    check_gene_filters("cpu")
