"""Modularity clustering of a neighbour graph — what the reference reaches through ``sc.tl.leiden``
(dance/modules/spatial/spatial_domain/spagcn.py:480-492 ``init="louvain"``, which really calls leiden;
dance/modules/single_modality/clustering/graphsc.py:568-587 ``run_leiden``).

scanpy / leidenalg / igraph are not installable here, and community detection is control flow on the host, off the
hot path.  This module optimises the SAME quality function leidenalg does for scanpy — Reichardt-Bornholdt modularity
with the configuration null model and a resolution parameter,

    Q = sum_c [ w_in(c) / m  -  resolution * (k(c) / (2 m))^2 ],

on the weighted UMAP connectivities — with the Louvain scheme (greedy local moves + graph aggregation) instead of
Leiden's refinement step.  Partitions are therefore of the same kind and quality but not label-for-label those of
leidenalg (which is itself seeded-stochastic; no reference output exists to pin — "parity unpinned").
Labels are numbered by decreasing community size, like scanpy's categorical.

The graph itself (exact kNN + fuzzy simplicial set) is built on the GPU by the same kernels as ``NeighborGraph``.
"""
import numpy as np
import scipy.sparse as sp


def _local_moves(indptr, indices, data, k, comm, resolution, two_m, rng):
    """One Louvain level: sweep the nodes (random order) until no move improves Q.  Returns True if anything moved."""
    n = len(k)
    tot = np.bincount(comm, weights=k, minlength=n).astype(np.float64)
    moved_any = False
    order = rng.permutation(n)
    while True:
        moved = 0
        for i in order:
            s, e = indptr[i], indptr[i + 1]
            if s == e:
                continue
            nb, w = indices[s:e], data[s:e]
            ci = comm[i]
            cs = comm[nb]
            # weight from i to every neighbouring community (self loops excluded)
            mask = nb != i
            uniq, inv = np.unique(cs[mask], return_inverse=True)
            if uniq.size == 0:
                continue
            k_in = np.bincount(inv, weights=w[mask])
            ki = k[i]
            tot[ci] -= ki
            gain = k_in - resolution * tot[uniq] * ki / two_m
            own = np.searchsorted(uniq, ci)
            stay = (k_in[own] if own < uniq.size and uniq[own] == ci else 0.0) - resolution * tot[ci] * ki / two_m
            j = int(np.argmax(gain))
            if gain[j] > stay + 1e-12 and uniq[j] != ci:
                comm[i] = uniq[j]
                tot[uniq[j]] += ki
                moved += 1
            else:
                tot[ci] += ki
        if moved == 0:
            break
        moved_any = True
    return moved_any


def louvain(adj, resolution: float = 1.0, random_state: int = 0, max_levels: int = 32) -> np.ndarray:
    """Community labels (int64 [n], 0 = largest community) of a symmetric weighted graph (scipy sparse)."""
    a = sp.csr_matrix(adj, dtype=np.float64)
    a = ((a + a.T) * 0.5).tocsr()  # symmetric by construction; harmless if it already is
    a.sort_indices()
    n = a.shape[0]
    rng = np.random.default_rng(random_state)
    two_m = float(a.sum())
    membership = np.arange(n)
    if two_m <= 0:
        return np.zeros(n, dtype=np.int64)
    for _ in range(max_levels):
        k = np.asarray(a.sum(1)).ravel()
        comm = np.arange(a.shape[0])
        if not _local_moves(a.indptr, a.indices, a.data, k, comm, resolution, two_m, rng):
            break
        _, comm = np.unique(comm, return_inverse=True)
        membership = comm[membership]
        nc = int(comm.max()) + 1
        s = sp.csr_matrix((np.ones(comm.size), (np.arange(comm.size), comm)), shape=(comm.size, nc))
        a = (s.T @ a @ s).tocsr()  # aggregated graph: intra-community weight becomes a self loop
        a.sort_indices()
        if nc == 1:
            break
    sizes = np.bincount(membership)
    rank = np.empty_like(sizes)
    rank[np.argsort(-sizes, kind="stable")] = np.arange(sizes.size)
    return rank[membership].astype(np.int64)


def modularity(adj, labels, resolution: float = 1.0) -> float:
    """Q of a partition (used by the tests: Louvain's result must beat trivial partitions)."""
    a = sp.csr_matrix(adj, dtype=np.float64)
    two_m = float(a.sum())
    labels = np.asarray(labels)
    k = np.asarray(a.sum(1)).ravel()
    coo = a.tocoo()
    w_in = coo.data[labels[coo.row] == labels[coo.col]].sum()
    tot = np.bincount(labels, weights=k)
    return float(w_in / two_m - resolution * ((tot / two_m)**2).sum())


def neighbors_connectivities(x, n_neighbors: int, device="cuda"):
    """``sc.pp.neighbors(adata, n_neighbors, use_rep="X").obsp["connectivities"]`` on the GPU: exact kNN (self is
    neighbour #0) + UMAP fuzzy simplicial set (dh_knn_bruteforce_f32, dh_umap_membership_f32, ...) -> scipy CSR."""
    import torch

    from .. import kernels
    xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    xt = xt.to(device=device, dtype=torch.float32).contiguous()
    n = xt.shape[0]
    idx, dist = kernels.knn(xt, min(int(n_neighbors), n))
    (rowptr, col, val), _ = kernels.umap_connectivities(idx, dist.contiguous())
    return sp.csr_matrix((val.cpu().numpy(), col.cpu().numpy(), rowptr.cpu().numpy()), shape=(n, n))


def leiden_like(x, n_neighbors: int, resolution: float = 1.0, random_state: int = 0, device="cuda") -> np.ndarray:
    """neighbours on the GPU + modularity clustering on the host: the stand-in for ``sc.pp.neighbors`` + ``sc.tl.leiden``."""
    return louvain(neighbors_connectivities(x, n_neighbors, device), resolution, random_state)
