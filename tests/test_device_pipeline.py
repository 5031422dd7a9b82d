"""The on-device preprocessing pipeline: normalisation -> scaling -> gene PCA / cell features -> CellFeatureGraph and
NeighborGraph with every intermediate kept on the device as a ``DeviceArray`` — ZERO device->host materialisations and ZERO
host->device uploads after the first one — and results equal to the host path's (same arithmetic, numpy slots).
CPU: kernel stand-ins (tests/cpu_ops.py) on CPU tensors; the GPU twin (test_gpu_transforms.py) runs the same checks on the kernels."""
import numpy as np
import pytest
import torch

import cpu_ops
from conftest import rel_err


@pytest.fixture
def cpu_kernels(monkeypatch):
    from dance_amd import kernels
    for name in cpu_ops.STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    return kernels


def _counts(n_cells, n_genes, seed):
    rng = np.random.default_rng(seed)
    x = ((rng.random((n_cells, n_genes)) < 0.3) * rng.integers(1, 9, (n_cells, n_genes))).astype(np.float32)
    x[np.arange(n_cells), rng.integers(0, n_genes, n_cells)] += 1  # no empty cell
    return x


def run_pipeline(device, device_in: bool):
    from dance_amd import data as dd
    from dance_amd.graph import LazyScipyCSR
    from dance_amd.transforms import CellPCA, Compose, WeightedFeaturePCA
    from dance_amd.transforms.graph import CellFeatureGraph, NeighborGraph
    from dance_amd.transforms.normalize import Log1P, NormalizeTotal
    x = _counts(240, 60, 0)
    slot = dd.DeviceArray(torch.from_numpy(x.copy()).to(device)) if device_in else x.copy()
    data = dd.Data(dd.AnnDataLite(slot), train_size=-1, val_size=0, test_size=0)
    data.set_config(feature_channel=None, feature_channel_type="X")
    uploads = {"n": 0}
    real = dd.to_device_matrix

    def counting(v, dev):
        if not isinstance(v, (dd.DeviceArray, torch.Tensor)):
            uploads["n"] += 1
        return real(v, dev)

    dd.to_device_matrix = counting
    import dance_amd.transforms.normalize as nz
    nz_real, nz.to_device_matrix = nz.to_device_matrix, counting
    copies0, lazy0 = dd.DeviceArray.host_copies, LazyScipyCSR.host_copies
    try:
        pipe = Compose(NormalizeTotal(target_sum=1e4, device=device), Log1P(device=device),
                       WeightedFeaturePCA(n_components=12, split_name="train", device=device),
                       CellFeatureGraph(cell_feature_channel="WeightedFeaturePCA", device=device),
                       CellPCA(n_components=8, device=device), NeighborGraph(n_neighbors=10, device=device))
        pipe(data)
        stats = dict(uploads=uploads["n"], host_copies=dd.DeviceArray.host_copies - copies0, lazy_graph_copies=LazyScipyCSR.host_copies - lazy0)
    finally:
        dd.to_device_matrix, nz.to_device_matrix = real, nz_real
    return data, stats


def check_pipeline(device):
    from dance_amd.data import DeviceArray
    from dance_amd.graph import LazyScipyCSR
    d_dev, st_dev = run_pipeline(device, device_in=True)
    assert st_dev == dict(uploads=0, host_copies=0, lazy_graph_copies=0), st_dev  # steady state: nothing crosses PCIe
    ad = d_dev.data
    assert all(isinstance(v, DeviceArray) for v in (ad.X, ad.obsm["WeightedFeaturePCA"], ad.varm["WeightedFeaturePCA"], ad.obsm["CellPCA"]))
    assert isinstance(ad.obsp["NeighborGraph"], LazyScipyCSR)
    d_host, st_host = run_pipeline(device, device_in=False)
    assert st_host["uploads"] == 1 and st_host["host_copies"] == 0  # a host matrix is uploaded once, by the first transform
    # same results whichever way the matrix came in
    assert np.array_equal(np.asarray(ad.X), np.asarray(d_host.data.X))
    for k in ("WeightedFeaturePCA", "CellPCA"):
        assert rel_err(np.asarray(ad.obsm[k]), np.asarray(d_host.data.obsm[k])) < 1e-6, k
    g_dev, g_host = ad.uns["CellFeatureGraph"], d_host.data.uns["CellFeatureGraph"]
    assert torch.equal(g_dev.rowptr, g_host.rowptr) and torch.equal(g_dev.col, g_host.col) and torch.equal(g_dev.val, g_host.val)
    assert torch.equal(g_dev.eid, g_host.eid)
    a, b = ad.obsp["NeighborGraph"], d_host.data.obsp["NeighborGraph"]
    assert np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices) and np.allclose(a.data, b.data, rtol=1e-6)
    # ... and the lazy host views behave as the arrays host code expects
    assert LazyScipyCSR.host_copies >= 1 and a.toarray().shape == (240, 240)
    x_host = d_dev.get_feature(channel_type="X", return_type="numpy", split_name="train")
    assert isinstance(x_host, np.ndarray) and x_host.shape == (240, 60) and float(ad.X.max()) == float(x_host.max())
    # the device normalisation equals the oracle restatement of scanpy's
    from oracle import normalize as on
    out = on.normalize_total(_counts(240, 60, 0), 1e4, exclude_highly_expressed=True, max_fraction=0.05)
    ref = np.log1p(out[0] if isinstance(out, tuple) else out)
    assert rel_err(np.asarray(ad.X), ref) < 1e-6


def test_on_device_pipeline_has_no_host_round_trips(cpu_kernels):
    check_pipeline("cpu")


def check_model_pipelines(device):
    """ScDSC.preprocessing_pipeline() and GraphSC.preprocessing_pipeline() end to end on DeviceArray input: the reference's step list
    (scdsc.py:113-138, graphsc.py:110-146) with nothing crossing PCIe between the steps, and shapes / invariants of what comes out."""
    import torch
    from dance_amd import data as dd
    from dance_amd.graph import LazyScipyCSR
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    rng = np.random.default_rng(3)
    lam = rng.gamma(0.5, 2.0, 300)
    x = rng.poisson(lam[None, :] * rng.uniform(0.3, 2.0, (400, 1))).astype(np.float32)
    x[7] = 0  # a cell without counts: filtered

    def run(pipe, slot):
        for t in pipe.transforms:
            if hasattr(t, "device"):
                t.device = device
            if hasattr(t, "pca_device"):
                t.pca_device = device
        data = dd.Data(dd.AnnDataLite(slot), train_size=-1, val_size=0, test_size=0)
        c0, l0 = dd.DeviceArray.host_copies, LazyScipyCSR.host_copies
        pipe(data)
        return data, dd.DeviceArray.host_copies - c0, LazyScipyCSR.host_copies - l0

    data, copies, lazy = run(ScDSC.preprocessing_pipeline(n_top_genes=60, n_neighbors=12), dd.DeviceArray(torch.from_numpy(x.copy()).to(device)))
    assert copies == 0 and lazy == 0
    ad = data.data
    assert ad.X.shape == (399, 60) and ad.raw.X.shape == (399, 60) and isinstance(ad.X, dd.DeviceArray) and isinstance(ad.raw.X, dd.DeviceArray)
    xs = ad.X.tensor
    assert float(xs.mean(0).abs().max()) < 1e-4 and float((xs.std(0, unbiased=True) - 1).abs().max()) < 1e-3   # sc.pp.scale
    # obs["n_counts"] is what the reference ends with: every in-place sc.pp.filter_cells overwrites it, so after the second cell filter
    # (scdsc.py:124) it holds the row sums of the log1p-normalised HVG matrix — the value ZINB's size factors are built from (:246)
    xf = x[np.arange(400) != 7].astype(np.float64)
    xf = xf[:, np.asarray(x.sum(0) >= 3)]
    tot = xf.sum(1)
    logn = np.log1p(xf / tot[:, None] * np.median(tot))
    kept = np.flatnonzero(np.asarray(x.sum(0) >= 3))                      # var names are the original column numbers
    hvg_cols = np.searchsorted(kept, np.asarray([int(v) for v in ad.var_names]))
    assert np.allclose(np.asarray(ad.obs["n_counts"]), logn[:, hvg_cols].sum(1), rtol=1e-4)
    assert not np.allclose(np.asarray(ad.obs["n_counts"]), tot, rtol=1e-3)   # not the raw totals NormalizeTotal recorded first
    g = ad.uns["NeighborGraph.hip"]
    assert g.n_rows == 399 and g.symmetric
    adj, xx, raw, n_counts = data.get_x()           # what ScDSC.fit receives (host views materialise here, once each)
    assert adj.shape == (399, 399) and xx.shape == raw.shape == (399, 60) and n_counts.shape == (399, )
    data, copies, lazy = run(GraphSC.preprocessing_pipeline(n_top_genes=80, n_components=10), dd.DeviceArray(torch.from_numpy(x.copy()).to(device)))
    assert copies == 0
    cg = data.data.uns["CellFeatureGraph"]
    n_cells, n_genes = data.data.X.shape
    assert (n_cells, n_genes) == (399, 80) and cg.number_of_nodes() == 399 + 80 and cg.ndata["features"].shape == (479, 10)
    assert np.allclose(data.data.X.tensor.sum(1).cpu().numpy(), 1.0, rtol=1e-5)     # normalize_total(target_sum=1) of the edge weights
    with pytest.raises(ValueError):
        GraphSC.preprocessing_pipeline(normalize_weights="bogus")

    # ---- scTAG (sctag.py:119-145): scDSC's matrix steps, then PCA on the device and the kNN graph in PCA space ------------------------
    from dance_amd.modules.single_modality.clustering.sctag import ScTAG
    data, copies, lazy = run(ScTAG.preprocessing_pipeline(n_top_genes=60, n_components=10, n_neighbors=12), dd.DeviceArray(torch.from_numpy(x.copy()).to(device)))
    assert copies == 0 and lazy == 0
    ad = data.data
    assert ad.X.shape == (399, 60) and ad.raw.X.shape == (399, 60) and isinstance(ad.obsm["CellPCA"], dd.DeviceArray) and ad.obsm["CellPCA"].shape == (399, 10)
    assert ad.uns["NeighborGraph.hip"].n_rows == 399 and "n_counts" in ad.obs
    adj, xx, raw, n_counts = data.get_x()
    assert adj.shape == (399, 399) and xx.shape == raw.shape == (399, 60) and n_counts.shape == (399, )

    # ---- SpaGCN (spagcn.py:715-731): gene-name filter, normalize_total(1e4), log1p, the two spatial graphs, PCA ------------------------
    import pandas as pd
    from dance_amd.modules.spatial.spatial_domain.spagcn import SpaGCN
    names = [f"G{i}" for i in range(300)]
    names[3], names[17], names[40] = "ERCC-0003", "MT-CO1", "mt-nd1"   # the third one survives: the match is case sensitive by default
    xy = rng.uniform(0, 50, (400, 2))
    adl = dd.AnnDataLite(dd.DeviceArray(torch.from_numpy(x.copy()).to(device)), var=pd.DataFrame(index=names),
                         obsm={"spatial": xy, "spatial_pixel": np.round(xy * 3).astype(np.int64)}, uns={"image": rng.integers(0, 255, (160, 160, 3)).astype(np.uint8)})
    pipe = SpaGCN.preprocessing_pipeline(dim=12)
    for t in pipe.transforms:
        if hasattr(t, "device"):
            t.device = device
    data = dd.Data(adl, train_size=-1, val_size=0, test_size=0)
    c0 = dd.DeviceArray.host_copies
    pipe(data)
    ad = data.data
    assert dd.DeviceArray.host_copies == c0 and ad.X.shape == (400, 298) and "ERCC-0003" not in ad.var.index and "mt-nd1" in ad.var.index
    rows = np.asarray(ad.X)  # (a host view, on request)
    keep = np.ones(300, dtype=bool)
    keep[[3, 17]] = False
    tot = x[:, keep].sum(1, keepdims=True)
    want = np.log1p(np.where(tot > 0, x[:, keep] / np.where(tot > 0, tot, 1) * 1e4, 0))
    assert rel_err(rows, want) < 1e-6
    assert ad.obsm["CellPCA"].shape == (400, 12) and ad.obsp["SpaGCNGraph"].shape == (400, 400) and ad.obsp["SpaGCNGraph2D"].shape == (400, 400)

    # ---- scHeteroNet (scheteronet.py:592-604): rare cell types dropped, filters, HVG, SaveRaw, normalize_total, size factors, log1p, graph ----
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import scHeteroNet
    lab = rng.integers(0, 4, 400)
    lab[:6] = 4                                            # a cell type with 6 <= 10 cells: removed
    one_hot = pd.DataFrame(np.eye(5, dtype=np.float32)[lab], columns=[f"t{i}" for i in range(5)], index=[str(i) for i in range(400)])
    adl = dd.AnnDataLite(dd.DeviceArray(torch.from_numpy(x.copy()).to(device)), obsm={"cell_type": one_hot})
    pipe = scHeteroNet.preprocessing_pipeline()
    for t in pipe.transforms:
        if hasattr(t, "device"):
            t.device = device
    pipe.transforms[3].n_top_genes = 70
    data = dd.Data(adl, train_size=-1, val_size=0, test_size=0)
    pipe(data)
    ad = data.data
    n_keep = 400 - 6 - (0 if lab[7] == 4 else 1)           # the rare type and the empty cell 7
    assert ad.X.shape == (n_keep, 70) and ad.raw.X.shape == (n_keep, 70) and ad.obsm["cell_type"].shape[0] == n_keep
    assert float(ad.obsm["cell_type"]["t4"].sum()) == 0
    sf = np.asarray(ad.obs["size_factors"])
    assert np.isclose(np.median(sf), 1.0) and np.allclose(np.asarray(ad.obs["n_counts"]) / np.median(np.asarray(ad.obs["n_counts"])), sf)
    assert ad.obsp["HeteronetGraph"].shape == (n_keep, n_keep) if "HeteronetGraph" in ad.obsp else True

    # ---- STAGATE (stagate.py:157-173): the dispersion flavours and the default seurat_v3 (loess) on the device --------------------------------
    from dance_amd.modules.spatial.spatial_domain.stagate import Stagate
    adl = dd.AnnDataLite(dd.DeviceArray(torch.from_numpy(x.copy()).to(device)), obsm={"spatial_pixel": np.round(xy * 3).astype(np.int64)})
    pipe = Stagate.preprocessing_pipeline(hvg_flavor="cell_ranger", n_top_hvgs=50, model_name="knn", n_neighbors=6)
    for t in pipe.transforms:
        if hasattr(t, "device"):
            t.device = device
    data = dd.Data(adl, train_size=-1, val_size=0, test_size=0)
    pipe(data)
    assert data.data.X.shape[0] == 400 and 50 <= data.data.X.shape[1] <= 60  # (ties at the cut-off are all kept, as scanpy does)
    assert data.data.obsp["StagateGraph"].shape == (400, 400)
    assert len(Stagate.preprocessing_pipeline().transforms) == 5 and len(Stagate.preprocessing_pipeline(hvg_flavor=None).transforms) == 4
    # the reference's default flavour works on counts: the same genes as the numpy restatement picks, exactly n_top_hvgs of them
    from oracle import normalize as on
    adl = dd.AnnDataLite(dd.DeviceArray(torch.from_numpy(x.copy()).to(device)), obsm={"spatial_pixel": np.round(xy * 3).astype(np.int64)})
    pipe = Stagate.preprocessing_pipeline(n_top_hvgs=50, model_name="knn", n_neighbors=6)
    assert type(pipe.transforms[0]).__name__ == "HighlyVariableGenesRawCount"
    for t in pipe.transforms:
        if hasattr(t, "device"):
            t.device = device
    data = dd.Data(adl, train_size=-1, val_size=0, test_size=0)
    before = dd.DeviceArray.host_copies
    pipe(data)
    assert dd.DeviceArray.host_copies == before
    want = on.highly_variable_genes_seurat_v3(x, n_top_genes=50)[0]
    assert data.data.X.shape == (400, 50) and data.data.var.index.tolist() == [str(i) for i in np.flatnonzero(want)]
    with pytest.raises(ValueError):
        Stagate.preprocessing_pipeline(hvg_flavor="bogus")


def test_model_pipelines_on_device_arrays(cpu_kernels):
    check_model_pipelines("cpu")
