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


def _umap_known_answer_case():
    """Inputs whose fuzzy simplicial set has a closed form under umap-learn's published algorithm (scanpy / umap-learn are not
    installable here, so these hand-computed values are the known-answer pin of row A11).  Three well-separated triples of
    collinear points at 0, a, a + b (k = 3, self first): for the outer points rho = d1 and
    1 + exp(-(d2 - d1) / sigma) = log2(3)  =>  sigma = (d2 - d1) / -ln(log2(3) - 1),  membership(2nd neighbour) = log2(3) - 1."""
    a, b = 1.0, 2.5
    pts = np.array([[o, 0.0] for base in (0.0, 100.0, 200.0) for o in (base, base + a, base + a + b)], dtype=np.float32)
    c = float(np.log2(3.0) - 1.0)
    return pts, a, b, c


def test_umap_closed_form_known_answers():
    pts, a, b, c = _umap_known_answer_case()
    idx, dist = og.knn_exact(pts, 3)
    sig, rho = og.smooth_knn_dist(dist, 3.0)
    # point 0 of a triple: neighbours at a, a + b; point 1: at a, b; point 2: at b, a + b
    for t in range(3):
        assert np.allclose(rho[3 * t:3 * t + 3], [a, a, b])
        want_sigma = np.array([b, b - a, a]) / -np.log(c)   # (d2 - d1) / -ln(log2 3 - 1)
        assert np.allclose(sig[3 * t:3 * t + 3], want_sigma, rtol=3e-5)  # bisection stops at |psum - log2 k| < 1e-5
    conn, _, _ = og.fuzzy_simplicial_set(idx, dist, 3)
    d = conn.toarray()
    for t in range(3):
        blk = d[3 * t:3 * t + 3, 3 * t:3 * t + 3]
        # directed memberships: nearest neighbour 1, second neighbour c; union a + b - ab
        w01 = 1 + 1 - 1          # 0 -> 1 nearest (1), 1 -> 0 nearest (1)
        w02 = c + c - c * c      # 0 -> 2 second (c), 2 -> 0 second (c)
        w12 = c + 1 - c * 1      # 1 -> 2 second (c), 2 -> 1 nearest (1)
        assert np.allclose(blk, [[0, w01, w02], [w01, 0, w12], [w02, w12, 0]], atol=2e-5)
    assert d.sum() == sum(d[3 * t:3 * t + 3, 3 * t:3 * t + 3].sum() for t in range(3))  # no edges between triples
    # k = 2: psum = 1 = log2(2) at the first iterate, so sigma = 1 and every kNN edge has membership exactly 1
    idx2, dist2 = og.knn_exact(pts, 2)
    sig2, _ = og.smooth_knn_dist(dist2, 2.0)
    conn2, _, _ = og.fuzzy_simplicial_set(idx2, dist2, 2)
    assert np.all(sig2 == 1.0) and np.all(conn2.data == 1.0)


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


# ---- pins: the oracle against outputs of the reference's OWN code (tests/golden/graph_builders.npz, produced by
# tests/golden/make_golden.py: methods lifted from /root/reference by AST and run on torch-CPU, DGL storage semantics
# supplied by oracle.ref_extract.DGLStubGraph) ---------------------------------------------------------------------
import os  # noqa: E402

import pytest  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "graph_builders.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("norm", [0, 1])
def test_cell_feature_graph_pinned_to_reference_code(gold, norm):
    ref = og.cell_feature_graph(gold["cfg_x"], normalize_edges=bool(norm))
    tag = f"cfg_norm{norm}_"
    assert np.array_equal(ref["src"], gold[tag + "src"]) and np.array_equal(ref["dst"], gold[tag + "dst"])  # edge order
    assert np.array_equal(ref["cell_id"], gold[tag + "cell_id"]) and np.array_equal(ref["feat_id"], gold[tag + "feat_id"])
    assert np.allclose(ref["weight"], gold[tag + "weight"], rtol=1e-6, atol=0)
    assert np.array_equal(gold[tag + "features"], np.vstack((gold["cfg_gene_feat"], gold["cfg_cell_feat"])))


def test_adaptive_sage_pinned_to_reference_code(gold):
    src, dst, w, cid = gold["sage_src"], gold["sage_dst"], gold["sage_w"], gold["sage_cid"]
    alpha, h, n_dst = gold["sage_alpha"], gold["sage_h"], int(gold["sage_n_dst"])
    n_genes = alpha.shape[0] - 2
    idx = osg.sage_alpha_index(cid[src], cid[:n_dst][dst], n_genes)
    m = h[src] * alpha[idx] * w[:, None]                                         # gnn.py:81-82
    assert np.allclose(m, gold["sage_m"], rtol=1e-6, atol=1e-7)
    neigh = osg.sage_neigh(src, dst, w, cid, cid[:n_dst], alpha, h, n_dst)       # + fn.mean, gnn.py:90
    assert np.allclose(neigh, gold["sage_neigh"], rtol=1e-5, atol=1e-6)
    assert np.all(gold["sage_neigh"][4] == 0)                                    # isolated destination -> 0
    # the layer output ignores neigh (gnn.py:92-96): act(Linear(h_dst))
    z = np.maximum(h[:n_dst] @ gold["sage_lin_w"].T + gold["sage_lin_b"], 0)
    assert np.allclose(z, gold["sage_z"], rtol=1e-5, atol=1e-6)


def test_heteronet_edges_pinned_to_reference_code(gold):
    assert np.array_equal(og.heteronet_edges(gold["het_feats"], knears=5), gold["het_edges"])


def test_spagcn_p_and_search_l_pinned_to_reference_code(gold):
    from oracle import spagcn as osp
    adj = gold["spa_adj"]
    # (python floats, as the reference is called: an np.float64 `l` would promote the float32 adj to float64)
    for l, p in gold["spa_p_at"]:
        assert osp.calculate_p(adj, float(l)) == pytest.approx(p, rel=1e-12)
    for p, l in gold["spa_search"]:
        got = osp.search_l(float(p), adj)
        assert (got is None and np.isnan(l)) or got == pytest.approx(l, rel=1e-12)


def test_graph_builder_golden_is_reproducible_here():
    """Where the reference tree exists (the build container) the committed vectors are regenerated and compared."""
    from oracle import ref_extract
    if not ref_extract.available():
        pytest.skip("reference tree not mounted (GPU box): committed golden vectors are used as they are")
    import importlib.util
    import tempfile

    import torch
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(GOLD), "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    with tempfile.TemporaryDirectory() as tmp:
        mg.HERE = tmp
        torch.manual_seed(0)
        mg.make_graph_builders()
        new, old = np.load(os.path.join(tmp, "graph_builders.npz")), np.load(GOLD)
        for k in old.files:
            if k.startswith("sage_lin") or k == "sage_z":
                continue  # nn.Linear init depends on the torch RNG stream of the generating run
            assert np.array_equal(new[k], old[k], equal_nan=True), k


@pytest.mark.parametrize("agg", ["sum", "mean"])
def test_weighted_graph_conv_pinned_to_reference_code(gold, agg):
    n_dst = int(gold["sage_n_dst"])
    got = osg.weighted_graph_conv(gold["wgc_src"], gold["wgc_dst"], gold["wgc_w"], gold["wgc_feat"], n_dst, gold["wgc_weight"],
                                  bias=gold["wgc_bias"], agg=agg, activation="relu")
    assert np.allclose(got, gold["wgc_out_" + agg], rtol=1e-5, atol=1e-6)


def test_spagcn_graph_xyz_pinned_to_reference_code(gold):
    assert int(gold["spg_dist_func_id"]) == 0  # euclidean
    xyz = og.spagcn_xyz(gold["spg_xy"], gold["spg_xy_pixel"], gold["spg_img"], float(gold["spg_alpha"]), int(gold["spg_beta"]))
    assert xyz.dtype == np.float32 and np.array_equal(xyz, gold["spg_xyz"])


def test_stagate_graphs_pinned_to_reference_code(gold):
    xy = gold["stg_xy"]
    assert np.array_equal(og.stagate_radius_graph(xy, radius=1.7).toarray(), gold["stg_radius"])
    assert np.array_equal(og.stagate_knn_graph(xy, n_neighbors=4).toarray(), gold["stg_knn"])


def test_refine_pinned_to_reference_code(gold):
    """The product's refine (host-side majority vote; no HIP involved) against the reference's pandas loop."""
    from dance_amd.modules.spatial.spatial_domain.spagcn import refine
    ids = [f"s{i}" for i in range(gold["ref_pred"].size)]
    for shape in ("hexagon", "square"):
        got = np.asarray(refine(ids, gold["ref_pred"], gold["ref_dis"], shape=shape), dtype=np.int64)
        assert np.array_equal(got, gold["ref_refined_" + shape])


def test_scdeepsort_predict_rule_matches_reference_method():
    """ScDeepSort.predict (argmax + "unsure" rule, scdeepsort.py:330-349): the reference's own method, lifted and run on a
    stand-in ``self``, against the mirrored class on the same probabilities (host-side logic only)."""
    from oracle import ref_extract
    if not ref_extract.available():
        pytest.skip("reference tree not mounted")
    import types
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    ref_predict = ref_extract.extract_method("dance/modules/single_modality/cell_type_annotation/scdeepsort.py", "ScDeepSort", "predict",
                                             {"dgl": types.SimpleNamespace(DGLGraph=object)})  # only a type annotation
    rng = np.random.default_rng(0)
    prob = rng.dirichlet(np.ones(5) * 0.7, size=200)
    prob[:20] = 0.2 + rng.normal(0, 1e-3, (20, 5))     # near-uniform rows: "unsure" at unsure_rate 2 means max < 0.4
    prob /= prob.sum(1, keepdims=True)
    stub = types.SimpleNamespace(predict_proba=lambda g: prob, num_labels=5)
    mine = object.__new__(ScDeepSort)
    mine.predict_proba, mine.num_labels = (lambda g: prob), 5
    for rate in (2.0, 1.0, 3.5):
        r_pred, r_unsure = ref_predict(stub, None, unsure_rate=rate, return_unsure=True)
        m_pred, m_unsure = mine.predict(None, unsure_rate=rate, return_unsure=True)
        assert np.array_equal(np.asarray(r_pred), np.asarray(m_pred)) and np.array_equal(np.asarray(r_unsure), np.asarray(m_unsure))
    assert np.array_equal(np.asarray(ref_predict(stub, None)), np.asarray(mine.predict(None)))


def test_graphsc_golden_regenerates_from_reference(tmp_path, monkeypatch):
    """graphsc.npz is what the reference's own GCNAE / GraphSC.fit produce NOW (build container only)."""
    import importlib.util
    import os

    import pytest

    from oracle import ref_extract
    if not ref_extract.available():
        pytest.skip("reference tree not present")
    here = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_gsc", os.path.join(here, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "HERE", str(tmp_path))
    mod.make_graphsc()
    new, old = np.load(tmp_path / "graphsc.npz"), np.load(os.path.join(here, "graphsc.npz"))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        if old[k].dtype.kind in "fc":
            assert np.allclose(new[k], old[k], rtol=1e-5, atol=1e-6), k
        else:
            assert np.array_equal(new[k], old[k]), k


@pytest.mark.parametrize("maker,fname", [("make_scheteronet", "scheteronet.npz"), ("make_scdsc_fit", "scdsc_fit.npz"), ("make_sctag", "sctag.npz"), ("make_stagate", "stagate.npz"), ("make_free_riders", "free_riders.npz"),
                                         ("make_gc_dec", "gc_dec.npz"), ("make_wgc_alpha", "wgc_alpha.npz"), ("make_small_transforms", "small_transforms.npz"),
                                         ("make_scheteronet_split", "scheteronet_split.npz"), ("make_gene_filters", "gene_filters.npz"),
                                         ("make_feature_feature_graph", "feature_feature_graph.npz"), ("make_graphsci", "graphsci.npz"),
                                         ("make_scdeepsort", "scdeepsort.npz")])
def test_model_goldens_regenerate_from_reference(tmp_path, monkeypatch, maker, fname):
    """Every model / transform golden is what the reference's own code produces NOW (build container only)."""
    import importlib.util
    import os

    from oracle import ref_extract
    if not ref_extract.available():
        pytest.skip("reference tree not present")
    here = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_golden_" + maker, os.path.join(here, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "HERE", str(tmp_path))
    getattr(mod, maker)()
    new, old = np.load(tmp_path / fname), np.load(os.path.join(here, fname))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        if old[k].dtype.kind in "fc":
            assert np.allclose(new[k], old[k], rtol=1e-4, atol=1e-5, equal_nan=True), k
        else:
            assert np.array_equal(new[k], old[k]), k
