"""Oracle for the graph builders of dance/transforms/graph (test infrastructure; see oracle/__init__.py).

PARITY UNPINNED by reference output: scanpy / umap-learn / dgl / numba cannot be installed here.  Restated from
the cited reference lines and from the published algorithms of the pinned dependencies ([3P-memory] where a
detail comes from the dependency, not from the reference tree).  scikit-learn IS importable and is used as a
cross-check of the exact-kNN definition in tests/test_oracle_graphs.py.
"""
import numpy as np
import scipy.sparse as sp

SMOOTH_K_TOLERANCE = 1e-5  # umap-learn umap_.py
MIN_K_DIST_SCALE = 1e-3


# ---- exact kNN -----------------------------------------------------------------------------------------------
def knn_exact(x, k, chunk=512, q_begin=0, q_end=None):
    """k nearest rows (self included) by (d2, index), d2 accumulated in f32 in feature order with separately
    rounded subtract / multiply / add — the bit-level definition shared with dh_knn_bruteforce_f32.
    Stands in for sklearn NearestNeighbors(n_neighbors=k).kneighbors (heteronet_graph.py:36-37,
    spatial_graph.py:147-149) and the exact branch of sc.pp.neighbors (neighbor_graph.py:52)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, d = x.shape
    kk = min(k, n)
    q_end = n if q_end is None else q_end
    idx = np.full((n, k), -1, dtype=np.int32)
    dist = np.full((n, k), np.inf, dtype=np.float32)
    for q0 in range(q_begin, q_end, chunk):
        q1 = min(q_end, q0 + chunk)
        acc = np.zeros((q1 - q0, n), dtype=np.float32)
        for t in range(d):
            diff = x[q0:q1, t, None] - x[None, :, t]
            acc = acc + diff * diff  # f32: rn(acc + rn(diff*diff))
        order = np.argsort(acc, axis=1, kind="stable")[:, :kk]  # stable: ties -> lower index
        idx[q0:q1, :kk] = order
        dist[q0:q1, :kk] = np.sqrt(np.take_along_axis(acc, order, axis=1))
    return idx[q_begin:q_end], dist[q_begin:q_end]


# ---- UMAP connectivities (umap-learn, as called by scanpy 1.10.1 sc.pp.neighbors(method="umap")) -------------
def smooth_knn_dist(distances, k, n_iter=64, local_connectivity=1.0, bandwidth=1.0):
    """umap_.smooth_knn_dist [3P-memory]: rho = first positive distance; sigma by bisection, double arithmetic,
    results stored as float32."""
    distances = np.asarray(distances, dtype=np.float32)
    n, kk = distances.shape
    target = np.log2(k) * bandwidth
    rho = np.zeros(n, dtype=np.float32)
    result = np.zeros(n, dtype=np.float32)
    mean_distances = float(np.mean(distances.astype(np.float64)))
    for i in range(n):
        lo, hi, mid = 0.0, np.inf, 1.0
        ith = distances[i]
        nz = ith[ith > 0.0]
        if nz.shape[0] >= local_connectivity:
            index = int(np.floor(local_connectivity))
            interpolation = local_connectivity - index
            if index > 0:
                rho[i] = nz[index - 1]
                if interpolation > SMOOTH_K_TOLERANCE:
                    rho[i] += interpolation * (nz[index] - nz[index - 1])
            else:
                rho[i] = interpolation * nz[0]
        elif nz.shape[0] > 0:
            rho[i] = np.max(nz)
        for _ in range(n_iter):
            psum = 0.0
            for j in range(1, kk):
                d = np.float32(ith[j] - rho[i])  # f32 - f32
                psum += np.exp(-(np.float64(d) / mid)) if d > 0 else 1.0
            if np.fabs(psum - target) < SMOOTH_K_TOLERANCE:
                break
            if psum > target:
                hi = mid
                mid = (lo + hi) / 2.0
            else:
                lo = mid
                mid = mid * 2 if hi == np.inf else (lo + hi) / 2.0
        result[i] = mid
        if rho[i] > 0.0:
            m = float(np.mean(ith.astype(np.float64)))
            if result[i] < MIN_K_DIST_SCALE * m:
                result[i] = MIN_K_DIST_SCALE * m
        elif result[i] < MIN_K_DIST_SCALE * mean_distances:
            result[i] = MIN_K_DIST_SCALE * mean_distances
    return result, rho


def compute_membership_strengths(knn_indices, knn_dists, sigmas, rhos):
    """umap_.compute_membership_strengths [3P-memory]: f32 arithmetic, self edges 0, d <= rho -> 1."""
    n, k = knn_indices.shape
    rows = np.repeat(np.arange(n, dtype=np.int32), k)
    cols = knn_indices.reshape(-1).astype(np.int32)
    d = (knn_dists.astype(np.float32) - rhos[:, None].astype(np.float32)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        vals = np.exp(-(d / sigmas[:, None].astype(np.float32))).astype(np.float32)
    vals[(d <= 0) | (sigmas[:, None] == 0)] = 1.0
    vals[knn_indices == np.arange(n)[:, None]] = 0.0
    vals = vals.reshape(-1)
    keep = cols >= 0
    return rows[keep], cols[keep], vals[keep]


def fuzzy_simplicial_set(knn_indices, knn_dists, n_neighbors):
    """umap_.fuzzy_simplicial_set with set_op_mix_ratio=1, local_connectivity=1 -> symmetric CSR (float32)."""
    n = knn_indices.shape[0]
    knn_dists = knn_dists.astype(np.float32)
    sigmas, rhos = smooth_knn_dist(knn_dists, float(n_neighbors))
    rows, cols, vals = compute_membership_strengths(knn_indices, knn_dists, sigmas, rhos)
    result = sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()
    result.eliminate_zeros()
    transpose = result.transpose().tocsr()
    prod = result.multiply(transpose)
    result = (result + transpose - prod).tocsr()  # f32 sparse arithmetic: (a + b) - a*b
    result.eliminate_zeros()
    result.sort_indices()
    return result.astype(np.float32), sigmas, rhos


def neighbor_graph(x, n_neighbors=15):
    """NeighborGraph.__call__ (neighbor_graph.py:50-57): connectivities of sc.pp.neighbors(knn=True,
    method='umap', metric='euclidean'), with the kNN stage exact (the reference's pynndescent branch at large N is
    approximate and seed-dependent; parity is defined against exact kNN)."""
    idx, dist = knn_exact(x, n_neighbors)
    conn, _, _ = fuzzy_simplicial_set(idx, dist, n_neighbors)
    return conn


# ---- HeteronetGraph (heteronet_graph.py:27-40) ---------------------------------------------------------------
def gauss_connectivities(knn_indices, knn_dists, knn=True):
    """sc.pp.neighbors(method="gauss") connectivities (neighbor_graph.py:37-39,52-55 forward ``method`` / ``knn``): scanpy 1.10.1
    ``Neighbors._compute_connectivities_diffmap`` [3P-memory], restated with explicit loops.  ``knn_indices`` / ``knn_dists``
    [n, k] with the point itself in column 0.
      knn=True : sigma_i^2 = median of the k - 1 squared neighbour distances; W_ij = sqrt(2 s_i s_j / (s_i^2 + s_j^2)) exp(-d_ij^2 /
                 (s_i^2 + s_j^2)) on the kNN entries, then W_ji = W_ij wherever i is not among j's neighbours (pattern = the union).
      knn=False: sigma_i^2 = (squared distance to the last of the k neighbours) / 4, W as above on ALL pairs from the dense distance
                 matrix (``knn_dists`` is then the dense [n, n] matrix and ``knn_indices`` its row-wise argsort), entries <= 1e-14
                 dropped."""
    n = knn_indices.shape[0]
    if not knn:
        d2 = np.asarray(knn_dists, dtype=np.float64)**2
        k = knn_indices.shape[1]
        sig2 = np.sort(d2, axis=1)[:, k - 1] / 4
        sig = np.sqrt(sig2)
        w = np.sqrt(2 * np.multiply.outer(sig, sig) / np.add.outer(sig2, sig2)) * np.exp(-d2 / np.add.outer(sig2, sig2))
        w[w <= 1e-14] = 0
        return sp.csr_matrix(w)
    ind = np.asarray(knn_indices)[:, 1:]
    d2 = np.asarray(knn_dists, dtype=np.float64)[:, 1:]**2
    sig2 = np.median(d2, axis=1)
    sig = np.sqrt(sig2)
    w = sp.lil_matrix((n, n), dtype=np.float64)
    for i in range(n):
        for t, j in enumerate(ind[i]):
            den = sig2[i] + sig2[j]
            w[i, j] = np.sqrt(2 * sig[i] * sig[j] / den) * np.exp(-d2[i, t] / den)
    for i in range(n):
        for j in ind[i]:
            if i not in set(ind[j]):
                w[j, i] = w[i, j]
    return w.tocsr()


def heteronet_edges(features, knears=5):
    """edge list [[i, j] for j in indices[i]] with indices = kneighbors of k+1 points INCLUDING self (:36-39)."""
    idx, _ = knn_exact(features, knears + 1)
    n = idx.shape[0]
    return np.stack([np.repeat(np.arange(n), idx.shape[1]), idx.reshape(-1)], axis=1).astype(np.int64)


# ---- StagateGraph (spatial_graph.py:143-151) -----------------------------------------------------------------
def stagate_knn_graph(xy, n_neighbors=5):
    """NearestNeighbors(n_neighbors).fit(xy).kneighbors_graph(xy): 0/1 CSR, self included (training-set query)."""
    idx, _ = knn_exact(xy, n_neighbors)
    n = idx.shape[0]
    a = sp.csr_matrix((np.ones(idx.size), (np.repeat(np.arange(n), idx.shape[1]), idx.reshape(-1))), shape=(n, n))
    a.sort_indices()
    return a


def stagate_radius_graph(xy, radius=1.0):
    """NearestNeighbors(radius).radius_neighbors_graph(xy): 0/1 CSR of pairs with distance <= radius (self incl.)."""
    xy = np.asarray(xy, dtype=np.float64)
    d2 = ((xy[:, None, :] - xy[None, :, :])**2).sum(-1)
    a = sp.csr_matrix((d2 <= radius * radius).astype(np.float64))
    a.sort_indices()
    return a


# ---- SpaGCNGraph (spatial_graph.py:36-62) --------------------------------------------------------------------
def spagcn_xyz(xy, xy_pixel, img, alpha, beta):
    """spatial_graph.py:40-58: per-spot patch mean -> variance-weighted grey -> z-score * max std(xy) * alpha."""
    g = np.zeros((xy.shape[0], 3))
    beta_half = round(beta / 2)
    x_lim, y_lim = img.shape[:2]
    for i, (x_pixel, y_pixel) in enumerate(xy_pixel):
        top = max(0, x_pixel - beta_half)
        left = max(0, y_pixel - beta_half)
        bottom = min(x_lim, x_pixel + beta_half + 1)
        right = min(y_lim, y_pixel + beta_half + 1)
        g[i] = np.mean(img[top:bottom, left:right], axis=(0, 1))
    g_var = g.var(0)
    z = (g * g_var).sum(1, keepdims=True) / g_var.sum()
    z = (z - z.mean()) / z.std()
    z *= xy.std(0).max() * alpha
    return np.hstack((xy, z)).astype(np.float32)


# ---- CellFeatureGraph (cell_feature_graph.py:34-79) ----------------------------------------------------------
def cell_feature_graph(feat, normalize_edges=True):
    """Edge list of the bipartite cell-gene graph in the REFERENCE edge order.

    Nodes: genes [0, G), cells [G, G+N).  Edges: nnz cell->gene (row-major nonzero order, :38-45), then nnz
    gene->cell, then one self loop per node with weight 1 (:69).  If normalize_edges, every node's in-edge
    weights are rescaled to in_deg * w / sum(w) BEFORE the self loops are added (:62-68).
    Returns dict(src, dst int64; weight f32 [E]; cell_id, feat_id int32 [G+N])."""
    feat = np.asarray(feat)
    n_cells, n_feats = feat.shape
    row, col = np.nonzero(feat)  # :38
    edata = np.asarray(feat[row, col], dtype=np.float32).ravel()  # :39
    row = row + n_feats  # :43
    src = np.hstack((row, col)).astype(np.int64)  # :44 (new `row`)
    dst = np.hstack((col, row)).astype(np.int64)  # :44 (new `col`)
    w = np.hstack((edata, edata)).astype(np.float32)  # :45
    n_nodes = n_cells + n_feats
    cell_id = np.concatenate((np.arange(n_feats, dtype=np.int32), -np.ones(n_cells, dtype=np.int32)))  # :56-57
    feat_id = np.concatenate((-np.ones(n_feats, dtype=np.int32), np.arange(n_cells, dtype=np.int32)))  # :58-59
    if normalize_edges:  # :62-68
        in_deg = np.bincount(dst, minlength=n_nodes)
        sums = np.bincount(dst, weights=w.astype(np.float64), minlength=n_nodes).astype(np.float32)
        w = ((in_deg[dst].astype(np.float32) * w) / sums[dst]).astype(np.float32)
    loops = np.arange(n_nodes, dtype=np.int64)  # :69
    return dict(src=np.concatenate((src, loops)), dst=np.concatenate((dst, loops)),
                weight=np.concatenate((w, np.ones(n_nodes, dtype=np.float32))), cell_id=cell_id, feat_id=feat_id,
                n_genes=n_feats, n_cells=n_cells)
