"""CPU: pin oracle.matrix against the reference's own known-answer tests (tests/utils/test_matrix.py:9-65)."""
import json
import os

import numpy as np

from conftest import GOLDEN
from oracle import matrix as om

KA = json.load(open(os.path.join(GOLDEN, "matrix_known_answers.json")))


def test_pairwise_distance_known_answers():
    mat = np.array(KA["mat"], dtype=np.float32)
    for fid, name in enumerate(("euclidean", "pearson", "spearman")):
        res = om.pairwise_distance(mat, fid)
        assert res.dtype == np.float32 and res.shape == (5, 5)
        assert np.allclose(np.array(KA[name]), res), name  # the reference test's own criterion (:53,:59,:65)


def test_normalize_known_answers():
    mat = np.array(KA["normalize_input"])
    exp = KA["normalize"]
    assert om.normalize(mat, mode="normalize", axis=0).tolist() == exp["normalize_axis0"]
    assert om.normalize(mat, mode="normalize", axis=1).tolist() == exp["normalize_axis1"]
    assert om.normalize(mat, mode="standardize", axis=0).tolist() == exp["standardize_axis0"]
    assert om.normalize(mat, mode="standardize", axis=1).tolist() == exp["standardize_axis1"]
    assert om.normalize(mat, mode="minmax", axis=0).tolist() == exp["minmax_axis0"]
    assert om.normalize(mat, mode="minmax", axis=1).tolist() == exp["minmax_axis1"]
    assert om.normalize(mat, mode="l2", axis=0).tolist() == (mat / np.sqrt((mat**2).sum(0))).tolist()
    assert om.normalize(mat, mode="l2", axis=1).tolist() == (mat / np.sqrt((mat**2).sum(1, keepdims=True))).tolist()


def test_mean_rank_matches_scipy():
    import scipy.stats
    rng = np.random.default_rng(0)
    for _ in range(5):
        x = rng.integers(0, 5, 12).astype(np.float32)
        assert np.array_equal(om.mean_rank_data(x), scipy.stats.rankdata(x).astype(np.float32))


def test_c_restatement_matches_numpy_oracle_and_golden():
    """oracle/csrc/pairwise_ref.c (the numba kernel restated in C) == the numpy restatement, bit for bit, and
    satisfies the reference's golden test; its kNN distance definition equals oracle.graphs.knn_exact's."""
    import pytest
    from oracle import cref
    from oracle import graphs as og
    if not cref.available():
        pytest.skip("oracle/_build not built (run __graft_entry__.build())")
    mat = np.array(KA["mat"], dtype=np.float32)
    assert np.allclose(np.array(KA["euclidean"]), cref.pairwise_euclidean(mat))
    x = np.random.default_rng(0).standard_normal((200, 7)).astype(np.float32) * 30
    assert np.array_equal(cref.pairwise_euclidean(x), om.pairwise_distance(x, 0))
    d2 = cref.sqdist(x, 10, 60)
    idx, dist = og.knn_exact(x, 9)
    order = np.argsort(d2, axis=1, kind="stable")[:, :9]
    assert np.array_equal(order, idx[10:60]) and np.array_equal(np.sqrt(np.take_along_axis(d2, order, 1)), dist[10:60])
