#!/usr/bin/env python
"""Reference-produced pin for NeighborGraph — run ONCE in any environment that has the reference's own
dependency stack (scanpy==1.10.1 with its umap-learn / pynndescent, as dance's install.sh pins it):

    python tests/golden/make_neighbor_graph_golden.py            # writes tests/golden/neighbor_graph.npz

What it records is exactly what ``dance/transforms/graph/neighbor_graph.py:50-57`` stores:
``sc.pp.neighbors(adata, use_rep=..., n_neighbors=k, metric=..., random_state=0, copy=True).obsp["connectivities"]`` (plus the
``distances`` matrix) for seeded inputs small enough that scanpy's neighbour search is EXACT (scanpy switches to approximate
NN-descent from 4096 points on; below that it is sklearn's brute force — the regime in which indices can be compared bit for bit
with the exact HIP search).  The image this repo is built in has neither scanpy nor umap-learn nor numba and no network, so the
fixture cannot be produced there; until it exists ``tests/test_gpu_graphs.py::test_neighbor_graph_vs_scanpy_fixture`` is an
expected failure with this file named as the remedy, and oracle/graphs.py stays "parity unpinned" for A11.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "neighbor_graph.npz")

# (name, cells, dims, k, metric): the reference's call sites — NeighborGraph() defaults on CellPCA (k = 15, euclidean, 50 PCs),
# scDSC's correlation graph on the scaled expression matrix (scdsc.py:128: k = 50, metric="correlation", channel="X"), scTAG's
# k = 15 on PCA (sctag.py:138), and a cosine case
CASES = [("euclid_k15", 3000, 50, 15, "euclidean"), ("corr_k50", 2000, 200, 50, "correlation"), ("cosine_k15", 1500, 32, 15, "cosine"),
         ("euclid_k5_small", 300, 10, 5, "euclidean")]


def inputs(name, n, d, seed):
    """Clustered gaussian points (float32), the same generator the GPU test re-creates."""
    rng = np.random.default_rng(seed)
    centres = rng.standard_normal((12, d)) * 3.0
    x = centres[rng.integers(0, 12, n)] + rng.standard_normal((n, d))
    return x.astype(np.float32)


def main():
    try:
        import anndata as ad
        import scanpy as sc
    except ImportError as e:  # pragma: no cover - only runs where the reference stack is absent
        sys.exit(f"scanpy / anndata are needed to produce the fixture ({e}); install the reference's pinned stack (dance install.sh) and re-run")
    out = {"scanpy_version": np.array(sc.__version__)}
    for i, (name, n, d, k, metric) in enumerate(CASES):
        x = inputs(name, n, d, 100 + i)
        adata = ad.AnnData(X=x.copy())
        adata.obsm["rep"] = x.copy()
        res = sc.pp.neighbors(adata, use_rep="rep", n_neighbors=k, knn=True, method="umap", metric=metric, random_state=0, copy=True)
        conn, dist = res.obsp["connectivities"].tocsr(), res.obsp["distances"].tocsr()
        conn.sort_indices()
        dist.sort_indices()
        out.update({f"{name}::x": x, f"{name}::k": np.array(k), f"{name}::metric": np.array(metric),
                    f"{name}::conn_indptr": conn.indptr.astype(np.int64), f"{name}::conn_indices": conn.indices.astype(np.int64),
                    f"{name}::conn_data": conn.data.astype(np.float32), f"{name}::dist_indptr": dist.indptr.astype(np.int64),
                    f"{name}::dist_indices": dist.indices.astype(np.int64), f"{name}::dist_data": dist.data.astype(np.float32)})
        print(f"{name}: n={n} d={d} k={k} {metric}: connectivities nnz={conn.nnz}")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
