"""CPU: sanity-pin the graph-builder oracle (no reference output exists for these — see oracle/__init__.py):
exact kNN against scikit-learn, UMAP connectivities against the algorithm's defining properties, the
CellFeatureGraph edge order against a hand-worked example of cell_feature_graph.py:38-69."""
import numpy as np
import scipy.sparse as sp
from sklearn.neighbors import NearestNeighbors

from oracle import graphs as og
from oracle import sage as osg


def test_knn_exact_matches_sklearn():
    rng = np.random.default_rng(0)
    for n, d, k in [(300, 10, 15), (257, 3, 6), (100, 70, 5)]:
        x = rng.standard_normal((n, d)).astype(np.float32)
        idx, dist = og.knn_exact(x, k)
        d2, i2 = NearestNeighbors(n_neighbors=k).fit(x).kneighbors(x)
        assert np.array_equal(idx, i2)
        assert np.allclose(dist, d2, rtol=1e-5, atol=1e-6)
        assert np.array_equal(idx[:, 0], np.arange(n)) and np.all(dist[:, 0] == 0)


def test_knn_ties_go_to_lower_index_and_short_input():
    x = np.zeros((6, 2), dtype=np.float32)
    x[3:] = 1
    idx, dist = og.knn_exact(x, 4)
    assert idx[0].tolist() == [0, 1, 2, 3] and idx[5].tolist() == [3, 4, 5, 0]
    idx, dist = og.knn_exact(x[:2], 4)
    assert idx[0].tolist() == [0, 1, -1, -1] and np.isinf(dist[0, 2])


def test_umap_connectivities_properties():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((400, 8)).astype(np.float32)
    k = 15
    idx, dist = og.knn_exact(x, k)
    sig, rho = og.smooth_knn_dist(dist, float(k))
    assert np.all(rho == dist[:, 1]) and np.all(sig > 0)
    # defining equation: sum_{j>=1} exp(-max(0, d - rho)/sigma) = log2(k)
    s = np.exp(-np.maximum(dist[:, 1:] - rho[:, None], 0) / sig[:, None]).sum(1)
    assert np.allclose(s, np.log2(k), atol=1e-3)
    conn, _, _ = og.fuzzy_simplicial_set(idx, dist, k)
    assert conn.dtype == np.float32 and abs(conn - conn.T).max() == 0
    assert conn.diagonal().sum() == 0 and conn.data.min() > 0 and conn.data.max() <= 1.0
    # pattern = union of kNN edges and their transposes (minus self)
    pat = sp.csr_matrix((np.ones(idx.size), (np.repeat(np.arange(400), k), idx.ravel())), shape=(400, 400))
    pat.setdiag(0)
    pat.eliminate_zeros()
    pat = ((pat + pat.T) > 0).astype(np.float32).tocsr()
    pat.sort_indices()
    assert np.array_equal(pat.indptr, conn.indptr) and np.array_equal(pat.indices, conn.indices)


def test_cell_feature_graph_reference_order():
    feat = np.array([[0, 2, 0], [1, 0, 3]], dtype=np.float32)  # 2 cells x 3 genes
    g = og.cell_feature_graph(feat, normalize_edges=False)
    # nonzeros row-major: (c0,g1)=2, (c1,g0)=1, (c1,g2)=3; cells are nodes 3,4
    assert g["src"].tolist() == [3, 4, 4, 1, 0, 2, 0, 1, 2, 3, 4]
    assert g["dst"].tolist() == [1, 0, 2, 3, 4, 4, 0, 1, 2, 3, 4]
    assert g["weight"].tolist() == [2, 1, 3, 2, 1, 3, 1, 1, 1, 1, 1]
    assert g["cell_id"].tolist() == [0, 1, 2, -1, -1] and g["feat_id"].tolist() == [-1, -1, -1, 0, 1]
    gn = og.cell_feature_graph(feat, normalize_edges=True)
    # cell 1 (node 4) has in-edges from g0 (1) and g2 (3): 2*1/4, 2*3/4; single in-edge nodes -> 1
    assert np.allclose(gn["weight"], [1, 1, 1, 1, 0.5, 1.5, 1, 1, 1, 1, 1])


def test_sage_alpha_index_and_mean():
    g = og.cell_feature_graph(np.array([[0, 2, 0], [1, 0, 3]], dtype=np.float32), normalize_edges=False)
    idx = osg.sage_alpha_index(g["cell_id"][g["src"]], g["cell_id"][g["dst"]], 3)
    # cell->gene: dst gene id; gene->cell: src gene id; gene loops: G=3; cell loops: G+1=4
    assert idx.tolist() == [1, 0, 2, 1, 0, 2, 3, 3, 3, 4, 4]
    alpha = np.arange(1, 6, dtype=np.float32)
    h = np.eye(5, dtype=np.float32)
    neigh = osg.sage_neigh(g["src"], g["dst"], g["weight"], g["cell_id"], g["cell_id"], alpha, h, 5)
    # node 4 (cell 1): edges from g0 (w=1, alpha[0]=1), g2 (w=3, alpha[2]=3), self (w=1, alpha[4]=5); mean of 3
    assert np.allclose(neigh[4], np.array([1, 0, 9, 0, 5]) / 3)
