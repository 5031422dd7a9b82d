"""Modularity clustering of a neighbour graph — what the reference reaches through ``sc.tl.leiden``
(dance/modules/spatial/spatial_domain/spagcn.py:480-492 ``init="louvain"``, which really calls leiden;
dance/modules/single_modality/clustering/graphsc.py:568-587 ``run_leiden``).

scanpy / leidenalg / igraph are not installable here, and community detection is control flow on the host, off the
hot path.  Both functions optimise the quality function leidenalg optimises for scanpy — Reichardt-Bornholdt modularity
with the configuration null model and a resolution parameter,

    Q = sum_c [ w_in(c) / m  -  resolution * (k(c) / (2 m))^2 ],

on the weighted UMAP connectivities: ``leiden`` with the Leiden scheme (fast local moving, refinement inside the
communities with randomised merges, aggregation of the REFINED partition — Traag, Waltman & van Eck 2019, the algorithm
leidenalg implements), ``louvain`` with the plain Louvain scheme it improves on.  Partitions are of leidenalg's kind and
carry its guarantee (every community is connected) but are not label-for-label its output: the algorithm is
seeded-stochastic and no reference output exists to pin — "parity unpinned", DESIGN.md.  Labels are numbered by decreasing
community size, like scanpy's categorical.

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


def _by_first_appearance(labels):
    """The same partition with labels 0, 1, ... in the order the communities first appear (a canonical form to compare two)."""
    _, first, inverse = np.unique(labels, return_index=True, return_inverse=True)
    return np.argsort(np.argsort(first))[inverse]


def _neighbour_weights(indptr, indices, data, v, comm, only=None):
    """[(community, weight between v and its members)] in community order, self loops excluded; ``only``: restrict the neighbours to
    those whose entry in ``only[0]`` equals ``only[1]`` (the refinement looks inside one parent community).  A python loop for the
    few neighbours of a kNN graph, array operations for the hundreds graph-sc asks for."""
    s, e = indptr[v], indptr[v + 1]
    if e - s <= 48:
        out = {}
        for j in range(s, e):
            u = indices[j]
            if u != v and (only is None or only[0][u] == only[1]):
                c = comm[u]
                out[c] = out.get(c, 0.0) + data[j]
        return sorted(out.items())
    nb, w = indices[s:e], data[s:e]
    keep = nb != v
    if only is not None:
        keep &= only[0][nb] == only[1]
    if not keep.any():
        return []
    uniq, inv = np.unique(comm[nb[keep]], return_inverse=True)
    return list(zip(uniq.tolist(), np.bincount(inv, weights=w[keep]).tolist()))


def _move_nodes_fast(indptr, indices, data, k, comm, resolution, two_m, rng):
    """Leiden's local moving: a queue of nodes in random order; a node goes to the community with the best gain (an empty one
    included), and only the neighbours it left behind are visited again.  Returns True if anything moved."""
    n = len(k)
    tot = np.bincount(comm, weights=k, minlength=n).astype(np.float64)
    size = np.bincount(comm, minlength=n)
    empty = [c for c in range(n) if size[c] == 0]
    queue = list(rng.permutation(n))
    queued = np.ones(n, dtype=bool)
    moved_any = False
    head = 0
    while head < len(queue):
        v = queue[head]
        head += 1
        queued[v] = False
        kv, cv = k[v], comm[v]
        w = _neighbour_weights(indptr, indices, data, v, comm)
        tot[cv] -= kv
        best_c, best = cv, dict(w).get(cv, 0.0) - resolution * tot[cv] * kv / two_m
        for c, wc in w:
            gain = wc - resolution * tot[c] * kv / two_m
            if gain > best + 1e-12:
                best_c, best = c, gain
        if best < -1e-12 and size[cv] > 1 and empty:        # alone is better than any neighbour (gain 0 in an empty community)
            best_c = empty.pop()
        if best_c != cv:
            comm[v] = best_c
            size[cv] -= 1
            size[best_c] += 1
            if size[cv] == 0:
                empty.append(cv)
            moved_any = True
            nb = indices[indptr[v]:indptr[v + 1]]
            again = nb[(comm[nb] != best_c) & ~queued[nb] & (nb != v)]
            if again.size:
                again = np.unique(again)
                queued[again] = True
                queue.extend(again.tolist())
        tot[comm[v]] += kv
    return moved_any


def _refine(indptr, indices, data, k, comm, resolution, two_m, rng, theta):
    """Leiden's refinement: inside every community start from singletons and merge well-connected nodes into well-connected
    sub-communities, chosen at random with probability ~ exp(gain / theta) among the non-negative gains.  Returns the refined
    membership (sub-communities never cross a community of ``comm``)."""
    n = len(k)
    refined = np.arange(n)
    r_tot = k.astype(np.float64).copy()          # degree sum of every refined community
    r_size = np.ones(n, dtype=np.int64)
    c_tot = np.bincount(comm, weights=k, minlength=n)
    # weight between a refined community and the rest of its parent community (for the well-connectedness test)
    rows = np.repeat(np.arange(n), np.diff(indptr))
    inside = (comm[rows] == comm[indices]) & (rows != indices)
    r_ext = np.bincount(rows[inside], weights=data[inside], minlength=n).astype(np.float64)
    node_ext = r_ext.copy()
    for v in rng.permutation(n):
        if r_size[refined[v]] != 1:
            continue                              # only nodes that are still alone are merged
        kv, parent = k[v], comm[v]
        if node_ext[v] < resolution * kv * (c_tot[parent] - kv) / two_m - 1e-12:
            continue                              # v is not well connected to its community
        w = dict(_neighbour_weights(indptr, indices, data, v, refined, only=(comm, parent)))
        own = refined[v]
        r_tot[own] -= kv
        cands, gains = [], []
        for c, wc in w.items():
            if c == own:
                continue
            if r_ext[c] < resolution * r_tot[c] * (c_tot[parent] - r_tot[c]) / two_m - 1e-12:
                continue                          # the target is not well connected
            gain = wc - resolution * r_tot[c] * kv / two_m
            if gain >= 0:
                cands.append(c)
                gains.append(gain)
        r_tot[own] += kv
        if not cands:
            continue
        gains = np.asarray(gains) / (two_m / 2)   # in units of Q, as the temperature theta is
        p = np.exp((gains - gains.max()) / theta)
        target = cands[int(rng.choice(len(cands), p=p / p.sum()))]
        # merge v into target: update sizes, degree sums and the external weights of the target
        r_ext[target] += node_ext[v] - 2 * w[target]
        r_tot[target] += kv
        r_size[target] += 1
        r_tot[own] -= kv
        r_size[own] = 0
        refined[v] = target
    return refined


def leiden(adj, resolution: float = 1.0, random_state: int = 0, theta: float = 0.01, n_iterations: int = -1, max_levels: int = 64) -> np.ndarray:
    """Community labels (int64 [n], 0 = largest community) of a symmetric weighted graph by the Leiden algorithm.
    ``n_iterations`` < 0 (leidenalg's and scanpy's default): repeat the whole algorithm, warm-started, until the partition stops changing."""
    a0 = sp.csr_matrix(adj, dtype=np.float64)
    a0 = ((a0 + a0.T) * 0.5).tocsr()
    a0.sort_indices()
    n = a0.shape[0]
    two_m = float(a0.sum())
    if two_m <= 0:
        return np.zeros(n, dtype=np.int64)
    rng = np.random.default_rng(random_state)
    membership = np.arange(n)
    it = 0
    while True:
        it += 1
        a, comm = a0, membership.copy()           # warm start: the previous result is the initial partition
        node_of = np.arange(n)                    # original node -> node of the current (aggregated) graph
        changed = False
        for _ in range(max_levels):
            k = np.asarray(a.sum(1)).ravel()
            moved = _move_nodes_fast(a.indptr, a.indices, a.data, k, comm, resolution, two_m, rng)
            changed |= moved
            _, comm = np.unique(comm, return_inverse=True)
            if int(comm.max()) + 1 == a.shape[0]:
                break                             # every community is a single node: nothing left to aggregate
            refined = _refine(a.indptr, a.indices, a.data, k, comm, resolution, two_m, rng, theta)
            _, refined = np.unique(refined, return_inverse=True)
            nr = int(refined.max()) + 1
            s = sp.csr_matrix((np.ones(refined.size), (np.arange(refined.size), refined)), shape=(refined.size, nr))
            a = (s.T @ a @ s).tocsr()
            a.sort_indices()
            parent = np.empty(nr, dtype=np.int64)
            parent[refined] = comm                # the aggregate starts from the NON-refined partition
            node_of = refined[node_of]
            comm = parent
            if nr == refined.size:
                break
        final = _by_first_appearance(comm[node_of])
        stable = not changed or np.array_equal(final, _by_first_appearance(membership))
        membership = final
        if (n_iterations < 0 and stable) or (n_iterations >= 0 and it >= n_iterations) or it >= 32:
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


def leiden_like(x, n_neighbors: int, resolution: float = 1.0, random_state: int = 0, device="cuda", method: str = "leiden") -> np.ndarray:
    """neighbours on the GPU + modularity clustering on the host: ``sc.pp.neighbors`` + ``sc.tl.leiden``."""
    if method not in ("leiden", "louvain"):
        raise ValueError(f"method must be 'leiden' or 'louvain', got {method!r}")
    cluster = leiden if method == "leiden" else louvain
    return cluster(neighbors_connectivities(x, n_neighbors, device), resolution, random_state)
