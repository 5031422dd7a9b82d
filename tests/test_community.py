"""CPU: the host modularity clustering behind SpaGCN's ``init="louvain"`` and graph-sc's ``run_leiden``
(dance_amd/utils/community.py; the reference calls scanpy's leiden: spagcn.py:480-492, graphsc.py:568-587)."""
import numpy as np
import scipy.sparse as sp

from dance_amd.utils.community import louvain, modularity


def _planted(n, c, seed, k=10):
    from sklearn.neighbors import kneighbors_graph
    rng = np.random.default_rng(seed)
    lab = np.sort(rng.integers(0, c, n))
    x = rng.standard_normal((n, 8)) + np.eye(c, 8)[lab] * 7
    a = kneighbors_graph(x, k, mode="connectivity")
    return ((a + a.T) > 0).astype(np.float32).tocsr(), lab


def test_louvain_recovers_planted_partition_and_orders_labels_by_size():
    from sklearn.metrics import adjusted_rand_score
    a, lab = _planted(900, 5, 0)
    got = louvain(a, 1.0)
    assert adjusted_rand_score(lab, got) > 0.98
    assert modularity(a, got) >= modularity(a, lab) - 1e-9
    sizes = np.bincount(got)
    assert np.all(np.diff(sizes) <= 0)  # 0 = largest community, like scanpy's categorical
    assert np.array_equal(got, louvain(a, 1.0))  # seeded: deterministic


def test_louvain_resolution_and_degenerate_graphs():
    a, lab = _planted(600, 4, 1)
    coarse, fine = louvain(a, 0.05), louvain(a, 5.0)
    assert len(np.unique(coarse)) <= len(np.unique(louvain(a, 1.0))) <= len(np.unique(fine))
    assert modularity(a, fine, 5.0) > modularity(a, coarse, 5.0)
    assert louvain(sp.csr_matrix((5, 5))).tolist() == [0] * 5  # no edges
    two = sp.csr_matrix(np.array([[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, 0, 1], [0, 0, 1, 0]], dtype=np.float32))
    got = louvain(two)
    assert got[0] == got[1] != got[2] == got[3]
