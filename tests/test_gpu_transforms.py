"""GPU: the transform drop-ins (dance_amd.transforms.graph) against the CPU oracle: bit-exact graph structure /
edge order, weights to 1e-5."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err
from oracle import graphs as og
from oracle import matrix as om

pytestmark = pytest.mark.gpu


def _counts(n_cells, n_genes, density, seed):
    rng = np.random.default_rng(seed)
    return ((rng.random((n_cells, n_genes)) < density) * rng.integers(1, 9, (n_cells, n_genes))).astype(np.float32)


def test_neighbor_graph_transform(cuda_device):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import NeighborGraph
    rng = np.random.default_rng(0)
    emb = (rng.standard_normal((6, 20))[rng.integers(0, 6, 900)] * 4 + rng.standard_normal((900, 20))).astype(np.float32)
    data = Data(AnnDataLite(np.zeros((900, 3), np.float32), obsm={"CellPCA": emb}))
    NeighborGraph(n_neighbors=15)(data)
    adj = data.data.obsp["NeighborGraph"]
    ref = og.neighbor_graph(emb, 15)
    assert sp.issparse(adj) and adj.dtype == np.float32 and adj.shape == (900, 900)
    assert np.array_equal(adj.indptr, ref.indptr) and np.array_equal(adj.indices, ref.indices)
    assert np.allclose(adj.data, ref.data, rtol=2e-5, atol=1e-7)
    assert abs(adj - adj.T).max() == 0
    dev_graph = data.data.uns["NeighborGraph.hip"]
    assert dev_graph.symmetric and dev_graph.nnz == adj.nnz
    # n_pcs picks leading columns, like scanpy's use_rep + n_pcs
    NeighborGraph(n_neighbors=10, n_pcs=5, out="g5")(data)
    ref5 = og.neighbor_graph(emb[:, :5], 10)
    assert np.array_equal(data.data.obsp["g5"].indices, ref5.indices)


def _cellgene_data(n_cells, n_genes, d, seed):
    from dance_amd.data import AnnDataLite, Data
    x = _counts(n_cells, n_genes, 0.2, seed)
    rng = np.random.default_rng(seed + 1)
    cell_feat = rng.standard_normal((n_cells, d)).astype(np.float32)
    gene_feat = rng.standard_normal((n_genes, d)).astype(np.float32)
    return x, cell_feat, gene_feat, Data(AnnDataLite(x, obsm={"feat": cell_feat}, varm={"feat": gene_feat}))


@pytest.mark.parametrize("normalize_edges", [True, False])
def test_cell_feature_graph_reference_edge_order(cuda_device, normalize_edges):
    from dance_amd.transforms.graph import CellFeatureGraph
    x, cell_feat, gene_feat, data = _cellgene_data(70, 40, 8, seed=3)
    CellFeatureGraph("feat", normalize_edges=normalize_edges)(data)
    g = data.data.uns["CellFeatureGraph"]
    ref = og.cell_feature_graph(x, normalize_edges=normalize_edges)
    src, dst = g.edges()
    assert np.array_equal(src.cpu().numpy(), ref["src"]) and np.array_equal(dst.cpu().numpy(), ref["dst"])  # bit-exact order
    w = g.edata["weight"].cpu().numpy()
    assert w.shape == (ref["weight"].size, 1) and rel_err(w[:, 0], ref["weight"]) < 1e-6
    assert np.array_equal(g.ndata["cell_id"].cpu().numpy(), ref["cell_id"])
    assert np.array_equal(g.ndata["feat_id"].cpu().numpy(), ref["feat_id"])
    assert np.array_equal(g.ndata["features"].cpu().numpy(), np.vstack((gene_feat, cell_feat)))
    n = 110
    assert g.number_of_nodes() == n and g.number_of_edges() == ref["src"].size
    assert np.array_equal(g.in_degrees().cpu().numpy(), np.bincount(ref["dst"], minlength=n))
    assert np.array_equal(g.out_degrees().cpu().numpy(), np.bincount(ref["src"], minlength=n))
    s, d, e = g.in_edges(45, form="all")
    sel = np.where(ref["dst"] == 45)[0]
    assert np.array_equal(np.sort(e.cpu().numpy()), sel) and np.array_equal(np.sort(s.cpu().numpy()), np.sort(ref["src"][sel]))


def test_cell_feature_graph_subgraph(cuda_device):
    from dance_amd.transforms.graph import CellFeatureGraph
    x, _, _, data = _cellgene_data(50, 30, 4, seed=9)
    CellFeatureGraph("feat")(data)
    g = data.data.uns["CellFeatureGraph"]
    ref = og.cell_feature_graph(x)
    nodes = np.concatenate((np.arange(30), 30 + np.array([4, 7, 8, 20, 33, 49])))  # genes + some cells (example :65)
    sub = g.subgraph(torch.from_numpy(nodes))
    lut = -np.ones(80, dtype=np.int64)
    lut[nodes] = np.arange(nodes.size)
    keep = (lut[ref["src"]] >= 0) & (lut[ref["dst"]] >= 0)
    s, d = sub.edges()
    assert np.array_equal(s.cpu().numpy(), lut[ref["src"][keep]]) and np.array_equal(d.cpu().numpy(), lut[ref["dst"][keep]])
    assert rel_err(sub.edata["weight"].cpu().numpy()[:, 0], ref["weight"][keep]) < 1e-6
    assert np.array_equal(sub.ndata["cell_id"].cpu().numpy(), ref["cell_id"][nodes])
    assert sub.number_of_nodes() == nodes.size


def test_pca_cell_feature_graph_pipeline(cuda_device):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import PCACellFeatureGraph
    x = _counts(120, 60, 0.3, 1)
    data = Data(AnnDataLite(x), train_size=90)
    data.set_config(feature_channel=None, feature_channel_type="X")
    PCACellFeatureGraph(n_components=16, split_name="train")(data)
    g = data.data.uns["CellFeatureGraph"]
    assert g.ndata["features"].shape == (180, 16) and g.number_of_edges() == 2 * int((x != 0).sum()) + 180


def test_heteronet_graph(cuda_device):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import HeteronetGraph
    rng = np.random.default_rng(5)
    x = rng.standard_normal((300, 80)).astype(np.float32)
    onehot = np.eye(4)[rng.integers(0, 4, 300)]
    data = Data(AnnDataLite(x, obsm={"cell_type": onehot}))
    t = HeteronetGraph(knn_num=5)
    edges = t.build_graph(x, knears=5)
    assert np.array_equal(edges, og.heteronet_edges(x, 5))
    t(data)
    g = data.data.uns["HeteronetGraph"]
    s, d = g.edges()
    assert np.array_equal(np.stack((s.cpu().numpy(), d.cpu().numpy()), 1), og.heteronet_edges(x, 5))
    assert np.array_equal(g.ndata["label"].cpu().numpy(), onehot.argmax(1))


def test_spagcn_and_stagate_graphs(cuda_device):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import SpaGCNGraph, SpaGCNGraph2D, StagateGraph
    rng = np.random.default_rng(2)
    n = 250
    xy = rng.integers(0, 60, (n, 2)).astype(np.float64)
    xy_pixel = (xy * 3 + rng.integers(0, 3, (n, 2))).astype(np.int64)
    img = rng.integers(0, 255, (200, 200, 3)).astype(np.float64)
    data = Data(AnnDataLite(np.zeros((n, 2), np.float32), obsm={"spatial": xy, "spatial_pixel": xy_pixel}, uns={"image": img}))
    SpaGCNGraph(alpha=1, beta=49)(data)
    ref = om.pairwise_distance(og.spagcn_xyz(xy, xy_pixel, img, 1, 49), 0)
    assert np.array_equal(data.data.obsp["SpaGCNGraph"], ref)
    SpaGCNGraph2D()(data)
    assert np.array_equal(data.data.obsp["SpaGCNGraph2D"], om.pairwise_distance(xy_pixel.astype(np.float32), 0))
    # distinct coordinates for kNN (ties are resolved by index in both implementations anyway)
    StagateGraph("knn", n_neighbors=6)(data)
    a, r = data.data.obsp["StagateGraph"], og.stagate_knn_graph(xy_pixel.astype(np.float32), 6)
    assert np.array_equal(a.indptr, r.indptr) and np.array_equal(a.indices, r.indices)
    StagateGraph("radius", radius=12.0, out="rad")(data)
    a, r = data.data.obsp["rad"].tocsr(), og.stagate_radius_graph(xy_pixel, 12.0)
    a.sort_indices()
    assert np.array_equal(a.indptr, r.indptr) and np.array_equal(a.indices, r.indices)


def test_device_pca_matches_sklearn_full(cuda_device):
    """dance_amd.utils.pca (opt-in ``device=`` of WeightedFeaturePCA / CellPCA) == scikit-learn's deterministic full-SVD
    PCA up to the per-component sign convention; both orientations (Gram of samples / covariance of features)."""
    from sklearn.decomposition import PCA
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms import CellPCA, WeightedFeaturePCA
    from dance_amd.utils.pca import pca_scores
    rng = np.random.default_rng(0)
    lat = rng.standard_normal((3000, 12)) @ rng.standard_normal((12, 200)) * 2 + rng.standard_normal((3000, 200)) * 0.3
    x = np.maximum(lat + 1.0, 0).astype(np.float32)   # cells x genes, non-negative like expression

    def same_up_to_sign(a, b, tol):
        s = np.sign((a * b).sum(0))
        assert rel_err(a * s, b) < tol

    for mat, k in ((x, 20), (np.ascontiguousarray(x.T), 20)):   # samples > features, samples < features
        ref = PCA(n_components=k, svd_solver="full").fit(mat)
        scores, comps, var = pca_scores(torch.from_numpy(mat).to(cuda_device), k)
        same_up_to_sign(scores.cpu().numpy(), ref.transform(mat), 2e-4)
        assert rel_err(var.cpu().numpy(), ref.explained_variance_) < 1e-4
        if comps is not None:
            same_up_to_sign(comps.cpu().numpy().T, ref.components_.T, 2e-4)
        # sign convention of scikit-learn 1.3 (the reference's pin): largest-magnitude entry of each U column positive
        u = scores.cpu().numpy()
        assert (u[np.abs(u).argmax(0), np.arange(k)] > 0).all()
    # through the transforms: device path vs the host path forced to the deterministic solver
    d_dev, d_host = Data(AnnDataLite(x.copy())), Data(AnnDataLite(x.copy()))
    WeightedFeaturePCA(16, device="cuda")(d_dev)
    gene_ref = PCA(n_components=16, svd_solver="full").fit_transform(x.T)
    same_up_to_sign(d_dev.data.varm["WeightedFeaturePCA"], gene_ref, 2e-4)
    cell_ref = (x / x.sum(1, keepdims=True)) @ d_dev.data.varm["WeightedFeaturePCA"]
    assert rel_err(d_dev.data.obsm["WeightedFeaturePCA"], cell_ref) < 1e-5
    CellPCA(10, device="cuda")(d_dev)
    CellPCA(10, svd_solver="full")(d_host)
    same_up_to_sign(d_dev.data.obsm["CellPCA"], d_host.data.obsm["CellPCA"], 2e-4)


def test_device_randomized_pca_matches_sklearn(cuda_device):
    """svd_solver="randomized" on the device (pca_scores_randomized: dh_gemm_f32 products, same numpy random stream as
    scikit-learn) == scikit-learn's randomised PCA with the same seed up to the component signs; both orientations and the
    two transforms (cell_feature.py:60-62 WeightedFeaturePCA, :176-181 CellPCA)."""
    from sklearn.decomposition import PCA
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms import CellPCA, WeightedFeaturePCA
    from dance_amd.utils.pca import pca_scores_randomized
    rng = np.random.default_rng(1)
    lat = rng.standard_normal((3000, 12)) @ rng.standard_normal((12, 200)) * 2 + rng.standard_normal((3000, 200)) * 0.3
    x = np.maximum(lat + 1.0, 0).astype(np.float32)

    def same_up_to_sign(a, b, tol):
        s = np.sign((a * b).sum(0))
        assert rel_err(a * s, b) < tol

    for mat, k in ((x, 12), (np.ascontiguousarray(x.T), 12), (x, 30)):   # 7 iterations (k < 0.1 min) and 4 (k = 30 of 200)
        ref = PCA(n_components=k, svd_solver="randomized", random_state=3)
        ref_scores = ref.fit_transform(mat)
        scores, comps, var = pca_scores_randomized(torch.from_numpy(mat).to(cuda_device), k, 3)
        kk = 12                                   # the 12 latent directions are separated; the noise tail is compared by variance
        same_up_to_sign(scores.cpu().numpy()[:, :kk], ref_scores[:, :kk], 5e-4)
        same_up_to_sign(comps.cpu().numpy().T[:, :kk], ref.components_.T[:, :kk], 5e-4)
        assert rel_err(var.cpu().numpy(), ref.explained_variance_) < 1e-4
        u = scores.cpu().numpy()
        assert (u[np.abs(u).argmax(0), np.arange(k)] > 0).all()
    d_dev, d_host = Data(AnnDataLite(x.copy())), Data(AnnDataLite(x.copy()))
    t = WeightedFeaturePCA(12, save_info=True, device="cuda")
    t.device_solver, t.device_random_state = "randomized", 5
    t(d_dev)
    host = PCA(n_components=12, svd_solver="randomized", random_state=5)
    gene_ref = host.fit_transform(x.T)
    same_up_to_sign(d_dev.data.varm["WeightedFeaturePCA"], gene_ref, 5e-4)
    assert rel_err(d_dev.data.uns["pca_explained_variance_ratio"], host.explained_variance_ratio_) < 1e-4
    assert rel_err(d_dev.data.uns["pca_mean"], host.mean_) < 1e-5
    c = CellPCA(10, svd_solver="randomized", device="cuda")
    c.device_random_state = 5
    c(d_dev)
    same_up_to_sign(d_dev.data.obsm["CellPCA"], PCA(n_components=10, svd_solver="randomized", random_state=5).fit_transform(x), 5e-4)
    with pytest.raises(ValueError):
        CellPCA(10, svd_solver="arpack", device="cuda")(d_host)


def test_on_device_pipeline_has_no_host_round_trips_gpu(cuda_device):
    """(f)3 on the kernels: NormalizeTotal -> Log1P -> WeightedFeaturePCA(device) -> CellFeatureGraph -> CellPCA(device) ->
    NeighborGraph with DeviceArray slots: zero uploads, zero device->host copies, results equal to the host-slot path
    (dh_dense_nnz_count_f32 / dh_dense_to_csr_f32 give the np.nonzero edge order)."""
    import test_device_pipeline as tdp
    tdp.check_pipeline("cuda")


def test_dense_to_csr_matches_numpy_nonzero(cuda_device):
    from dance_amd import kernels
    rng = np.random.default_rng(5)
    for n, m, dens in ((1, 1, 1.0), (7, 63, 0.5), (130, 64, 0.1), (257, 2000, 0.1), (33, 129, 0.0)):
        x = ((rng.random((n, m)) < dens) * rng.standard_normal((n, m))).astype(np.float32)
        xt = torch.from_numpy(x).to(cuda_device)
        rp, col, val = kernels.dense_to_csr(xt)
        r, c = np.nonzero(x)
        assert np.array_equal(col.cpu().numpy(), c) and np.array_equal(val.cpu().numpy(), x[r, c])
        assert np.array_equal(rp.cpu().numpy(), np.concatenate(([0], np.cumsum(np.bincount(r, minlength=n)))))
        # strided rows (a column window of a wider matrix)
        if m > 8:
            rp2, col2, val2 = kernels.dense_to_csr(xt[:, 3:m - 2])
            r2, c2 = np.nonzero(x[:, 3:m - 2])
            assert np.array_equal(col2.cpu().numpy(), c2) and np.array_equal(val2.cpu().numpy(), x[:, 3:m - 2][r2, c2])


def test_filter_and_hvg_on_device(cuda_device):
    import test_filter_transforms as tft
    tft.check_filters("cuda")


def test_feature_feature_graph_on_device(cuda_device):
    import test_feature_feature_graph as tffg
    tffg.check_feature_feature_graph("cuda")


def test_gene_filters_on_device(cuda_device):
    import test_gene_filters as tgf
    tgf.check_gene_filters("cuda")


def test_hvg_batch_key_on_device(cuda_device):
    import test_filter_transforms as tft
    tft.check_hvg_batches("cuda")


def test_seurat_v3_hvg_on_device(cuda_device):
    import test_filter_transforms as tft
    tft.check_seurat_v3("cuda")


def test_model_pipelines_on_device_arrays_gpu(cuda_device):
    import test_device_pipeline as tdp
    tdp.check_model_pipelines("cuda")
