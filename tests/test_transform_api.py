"""CPU: the host-side mirror keeps the reference's operator API — reprs pinned by the reference's
tests/transforms/test_basics.py:5-30, constructor signatures of SURVEY.md §8b.1, registry scopes, Data accessors."""
import inspect

import numpy as np
import pytest

from dance_amd.data import AnnDataLite, Data
from dance_amd.registry import REGISTRY, resolve_from_registry
from dance_amd.transforms import CellPCA, Compose, SetConfig, WeightedFeaturePCA
from dance_amd.transforms.graph import (CellFeatureGraph, HeteronetGraph, NeighborGraph, PCACellFeatureGraph, SpaGCNGraph,
                                        SpaGCNGraph2D, StagateGraph)


def test_reprs_match_reference_test_basics():
    assert repr(CellPCA(n_components=100)) == "CellPCA(n_components=100)"
    assert repr(WeightedFeaturePCA(n_components=100, split_name="train")) == (
        "WeightedFeaturePCA(n_components=100, split_name='train', feat_norm_mode=None, feat_norm_axis=0)")
    t = NeighborGraph(n_neighbors=10, n_pcs=None, knn=True, random_state=0, method="umap", metric="euclidean")
    assert repr(t) == ("NeighborGraph(n_neighbors=10, n_pcs=None, knn=True, random_state=0, method='umap', "
                       "metric='euclidean')")
    assert repr(PCACellFeatureGraph(n_components=100, split_name="train")) == "PCACellFeatureGraph(n_components=100, split_name='train')"
    assert repr(SpaGCNGraph(alpha=1, beta=2)) == "SpaGCNGraph(alpha=1, beta=2)"
    assert len(t.hexdigest()) == 32 and t.out == "NeighborGraph"


def test_constructor_signatures():
    def params(cls):
        return {k: v.default for k, v in inspect.signature(cls.__init__).parameters.items() if k not in ("self", "kwargs", "device")}

    assert params(NeighborGraph) == dict(n_neighbors=15, n_pcs=None, knn=True, random_state=0, method="umap",
                                         metric="euclidean", channel="CellPCA")
    p = params(CellFeatureGraph)
    assert p["gene_feature_channel"] is None and p["mod"] is None and p["normalize_edges"] is True
    assert params(PCACellFeatureGraph) == dict(n_components=400, split_name=None, normalize_edges=True, feat_norm_mode=None,
                                               feat_norm_axis=0, mod=None, log_level="WARNING")
    assert params(HeteronetGraph) == dict(knn_num=5, distance_metrics="l2", random_state=0, channel=None, channel_type="X",
                                          ignore_first=False)
    assert params(StagateGraph) == dict(model_name="radius", radius=1, n_neighbors=5, channel="spatial_pixel", channel_type="obsm")
    assert params(SpaGCNGraph2D) == dict(channel="spatial_pixel")
    with pytest.raises(ValueError):
        StagateGraph("bogus")  # spatial_graph.py:135-136


def test_registry_scopes():
    for name in ("CellFeatureGraph", "PCACellFeatureGraph", "NeighborGraph", "HeteronetGraph"):
        assert resolve_from_registry(name, "preprocessor.graph.cell").__name__ == name
    for name in ("SpaGCNGraph", "SpaGCNGraph2D", "StagateGraph"):
        assert resolve_from_registry(name, "preprocessor.graph.spatial").__name__ == name
    with pytest.raises(KeyError):
        resolve_from_registry("Nope", "preprocessor.graph.cell")
    assert not REGISTRY.is_leaf_node("preprocessor.graph")


def test_data_container():
    x = np.arange(20, dtype=np.float32).reshape(5, 4)
    d = Data(AnnDataLite(x, obsm={"emb": x[:, :2]}), train_size=3, val_size=0, test_size=-1)
    assert d.train_idx == [0, 1, 2] and d.test_idx == [3, 4] and d.val_idx is None
    assert d.num_cells == 5 and d.num_features == 4
    d.set_config(feature_channel="emb", feature_channel_type="obsm")
    assert d.get_x("train").shape == (3, 2)
    assert d.get_feature(channel_type="X", return_type="sparse").nnz == 19
    assert d.get_feature(channel="emb", return_type="torch").shape == (5, 2)
    with pytest.raises(KeyError):
        d.set_config(bogus=1)
    with pytest.raises(KeyError):
        d.set_config(feature_channel="other")
    d.set_config(feature_channel="other", overwrite=True)
    Compose(SetConfig({"feature_channel": "emb", "feature_channel_type": "obsm"}))(d)
    assert d.config["feature_channel"] == "emb"
    with pytest.raises(ValueError):
        Data(AnnDataLite(x), train_size=9)


def test_weighted_feature_pca_shapes():
    rng = np.random.default_rng(0)
    x = rng.poisson(1.0, (40, 30)).astype(np.float32)
    d = Data(AnnDataLite(x), train_size=30)
    d.set_config(feature_channel=None, feature_channel_type="X")
    WeightedFeaturePCA(n_components=8, split_name="train")(d)
    assert d.data.obsm["WeightedFeaturePCA"].shape == (40, 8) and d.data.varm["WeightedFeaturePCA"].shape == (30, 8)
    assert d.data.obsm["WeightedFeaturePCA"].dtype == np.float32


def test_block_sampler_on_cpu_tensors(monkeypatch):
    """The DGL-like graph / block / sampler / loader surface around the block builder: lazy src/dst frames, seeds-first
    source order, batching.  The builder itself is a HIP kernel (dh_block_plan / dh_block_fill, GPU-tested against the same
    torch-op restatement); here that restatement is injected so the host logic runs on CPU tensors."""
    import torch

    import cpu_ops
    from dance_amd import kernels
    from dance_amd.cellgraph import CellGeneGraph, DataLoader, NeighborSampler
    monkeypatch.setattr(kernels, "block_build", cpu_ops.block_build)
    # 3 genes (0..2), 4 cells (3..6); CSR by destination with a self loop per node
    src = [[0, 3, 5], [1, 4], [2, 3, 6], [0, 2, 3], [1, 4], [0, 5], [2, 6]]
    rowptr = torch.tensor(np.cumsum([0] + [len(r) for r in src]), dtype=torch.int32)
    col = torch.tensor([c for r in src for c in r], dtype=torch.int32)
    g = CellGeneGraph(rowptr, col, torch.arange(col.numel(), dtype=torch.float32), None, 7,
                      {"cell_id": torch.tensor([0, 1, 2, -1, -1, -1, -1], dtype=torch.int32), "features": torch.arange(14.).reshape(7, 2)})
    _, out_nodes, (blk, ) = NeighborSampler([-1]).sample(g, torch.tensor([5, 3]))
    assert blk.number_of_dst_nodes() == 2 and blk.number_of_src_nodes() == 4
    assert blk.srcdata["_ID"].tolist() == [5, 3, 0, 2]          # seeds first, then remaining in-neighbours ascending
    assert blk.dstdata["_ID"].tolist() == [5, 3]
    assert blk.rowptr.tolist() == [0, 2, 5] and blk.col.tolist() == [2, 0, 2, 3, 1]   # local source ids
    assert "features" in blk.srcdata and "nope" not in blk.srcdata
    assert dict.__len__(blk.srcdata) == 1                       # nothing but _ID gathered yet
    assert torch.equal(blk.srcdata["features"], g.ndata["features"][[5, 3, 0, 2]])
    assert torch.equal(blk.dstdata["cell_id"], torch.tensor([-1, -1], dtype=torch.int32))
    assert set(blk.srcdata.keys()) == {"_ID", "cell_id", "features"}
    blk.srcdata["h"] = torch.zeros(4, 1)
    assert "h" in dict(blk.srcdata.items())
    with pytest.raises(KeyError):
        blk.srcdata["nope"]
    batches = [o.tolist() for _, o, _ in DataLoader(g, torch.tensor([3, 4, 5, 6]), NeighborSampler([-1]), batch_size=3)]
    assert batches == [[3, 4, 5], [6]]
    shuffled = [o for _, o, _ in DataLoader(g, torch.tensor([3, 4, 5, 6]), NeighborSampler([-1]), batch_size=4, shuffle=True,
                                            generator=torch.Generator().manual_seed(0))]
    assert sorted(shuffled[0].tolist()) == [3, 4, 5, 6]


def test_as_graph_cache_is_identity_keyed(monkeypatch):
    """a WeakKeyDictionary keyed by sparse tensors compared keys with ``==`` (aten::eq is not implemented for
    sparse layouts) and raised on the second call with the same adjacency — the reference's epoch loop passes the same
    ``adj`` every epoch (scdsc.py:257-288)."""
    import gc

    import torch

    from dance_amd import graph
    built = []
    monkeypatch.setattr(graph.CSRGraph, "from_torch_sparse", classmethod(lambda cls, adj, device=None: (built.append(1), object())[1]))
    adj = torch.sparse_coo_tensor(torch.tensor([[0, 1], [1, 0]]), torch.ones(2), (2, 2))
    other = torch.sparse_coo_tensor(torch.tensor([[0, 1], [1, 0]]), torch.ones(2), (2, 2))
    g1, g2, g3 = graph.as_graph(adj), graph.as_graph(adj), graph.as_graph(other)
    assert g1 is g2 and g3 is not g1 and len(built) == 2
    del adj, other
    gc.collect()
    assert len(graph._CACHE) == 0  # entries die with their tensors


def test_randomized_pca_host_logic_matches_sklearn():
    """dance_amd.utils.pca.pca_scores_randomized (device solver for sklearn's ``svd_solver="randomized"``, the solver
    ``"auto"`` picks for the reference's WeightedFeaturePCA / CellPCA shapes) with the GEMMs injected as torch-CPU
    stand-ins: same numpy random stream, same iteration count, same subspace as scikit-learn's randomized_svd -> equal
    scores / components / variances to fp32 rounding, up to the per-component sign (scikit-learn >= 1.5 decides it on
    the feature side, 1.3 — the reference's pin — on the sample side, which is the rule implemented and checked here)."""
    import torch
    from sklearn.decomposition import PCA

    import cpu_ops
    from dance_amd.utils.pca import pca_scores_randomized
    rng = np.random.RandomState(0)
    for n, f, k in ((300, 40, 8), (40, 500, 6), (600, 100, 30), (64, 64, 5)):   # tall, wide, k >= 0.1 min (4 its), square
        x = (rng.randn(n, 12) @ rng.randn(12, f) * np.linspace(3, 1, f) + 0.3 * rng.randn(n, f)).astype(np.float32)
        ref = PCA(n_components=k, svd_solver="randomized", random_state=7)
        ref_scores = ref.fit_transform(x)
        scores, comps, var = (t.numpy() for t in pca_scores_randomized(torch.from_numpy(x), k, 7, ops=cpu_ops))
        sg = np.sign((scores * ref_scores).sum(0))
        assert np.abs(scores * sg - ref_scores).max() < 5e-5 * np.abs(ref_scores).max()
        assert np.abs(comps * sg[:, None] - ref.components_).max() < 5e-4
        assert np.abs(var - ref.explained_variance_).max() < 1e-5 * ref.explained_variance_.max()
        assert (scores[np.abs(scores).argmax(0), np.arange(k)] > 0).all()      # svd_flip, u-based (scikit-learn 1.3)
        # a RandomState instance is consumed exactly like the seed
        again = pca_scores_randomized(torch.from_numpy(x), k, np.random.RandomState(7), ops=cpu_ops)[0].numpy()
        assert np.array_equal(again, scores)
    with pytest.raises(ValueError):
        pca_scores_randomized(torch.zeros(5, 4), 5, 0, ops=cpu_ops)
    with pytest.raises(TypeError):
        pca_scores_randomized(torch.zeros(5, 4, dtype=torch.float64), 2, 0, ops=cpu_ops)


def _scale_in_place(adata, factor=1.0):
    adata.X = adata.X * factor


def test_anndata_transform_interface():
    """The cases of the reference's tests/transforms/test_interface.py:22-50 with a local in-place function in scanpy's place:
    a callable and its dotted path do the same thing; a module is not callable; an unknown attribute does not resolve."""
    from dance_amd.transforms import AnnDataTransform, Compose
    x = np.random.default_rng(123).random((5, 3)).astype(np.float32)
    a, b = Data(AnnDataLite(x.copy())), Data(AnnDataLite(x.copy()))
    AnnDataTransform(_scale_in_place, factor=100)(a)
    AnnDataTransform("test_transform_api._scale_in_place", factor=100)(b)
    assert a.data.X.tolist() == b.data.X.tolist() == (x * 100).tolist()
    t = AnnDataTransform("numpy.log1p")
    assert repr(t) == "AnnDataTransform(func=numpy.log1p, func_kwargs={})"
    with pytest.raises(TypeError):
        AnnDataTransform(np)
    with pytest.raises(TypeError):
        AnnDataTransform("numpy.linalg")
    with pytest.raises(AttributeError):
        AnnDataTransform("numpy.dosenot_exist")
    Compose(AnnDataTransform(_scale_in_place, factor=0.5), AnnDataTransform(_scale_in_place, factor=4))(a)
    assert np.allclose(a.data.X, x * 200)
    assert resolve_from_registry("AnnDataTransform", "preprocessor.interface") is AnnDataTransform


def test_utils_seed_and_device():
    import random

    import torch
    from dance_amd.utils import default, get_device, hexdigest, set_seed
    assert get_device("auto") == "cuda" and get_device("cuda:1") == "cuda:1"
    assert default(None, 3) == 3 and default(0, 3) == 0 and hexdigest("abc") == "900150983cd24fb0d6963f7d28e17f72"
    set_seed(5, cuda=False)
    a = (random.random(), float(np.random.rand()), float(torch.rand(1)))
    set_seed(5, cuda=False)
    assert a == (random.random(), float(np.random.rand()), float(torch.rand(1)))
