"""Generate the committed golden vectors from the REFERENCE's own code (run in the build container only).

    python tests/golden/make_golden.py

* gcn_layers.npz — the reference's GNNLayer (dance/modules/single_modality/clustering/scdsc.py:475-501) and
  GraphConvolution (dance/modules/spatial/spatial_domain/spagcn.py:337-366), AST-extracted from
  /root/reference and executed on torch-CPU: forward outputs and autograd gradients on seeded inputs.
* matrix_known_answers.json — the known-answer vectors of the reference's tests/utils/test_matrix.py:9-65
  (input matrix :33-39; expected values recomputed exactly as that test does, with scipy).

The reference's DGL / scanpy / numba call sites cannot run here (packages absent), so no golden file exists
for them; see oracle/__init__.py.
"""
import itertools
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.spatial.distance
import scipy.stats
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_extract  # noqa: E402
from oracle.layers import scipy_to_torch_coo  # noqa: E402


def make_gcn_layers():
    rng = np.random.default_rng(20240925)
    n, fin, fout = 96, 40, 24
    x = rng.standard_normal((n, fin)).astype(np.float32)
    w = (rng.standard_normal((fin, fout)) * 0.2).astype(np.float32)
    b = (rng.standard_normal(fout) * 0.1).astype(np.float32)
    dy = rng.standard_normal((n, fout)).astype(np.float32)
    adj = sp.random(n, n, density=0.08, random_state=7, format="csr", dtype=np.float32)
    adj.data = rng.uniform(0.1, 1.0, adj.nnz).astype(np.float32)
    adj_dense = np.exp(-rng.uniform(0, 3, (n, n))).astype(np.float32)  # SpaGCN-style dense kernel
    out = dict(x=x, w=w, b=b, dy=dy, adj_indptr=adj.indptr.astype(np.int32), adj_indices=adj.indices.astype(np.int32),
               adj_data=adj.data, adj_dense=adj_dense)

    GNNLayer = ref_extract.extract("dance/modules/single_modality/clustering/scdsc.py", "GNNLayer")
    GraphConvolution = ref_extract.extract("dance/modules/spatial/spatial_domain/spagcn.py", "GraphConvolution")
    a_coo = scipy_to_torch_coo(adj)

    for active in (True, False):
        layer = GNNLayer(fin, fout)
        layer.weight.data = torch.from_numpy(w.copy())
        xt = torch.from_numpy(x.copy()).requires_grad_(True)
        y = layer(xt, a_coo, active=active)
        y.backward(torch.from_numpy(dy))
        tag = "gnn_act" if active else "gnn_lin"
        out[f"{tag}_out"] = y.detach().numpy()
        out[f"{tag}_dW"] = layer.weight.grad.numpy().copy()
        out[f"{tag}_dX"] = xt.grad.numpy().copy()

    for tag, a in (("gc_sparse", a_coo), ("gc_dense", torch.from_numpy(adj_dense))):
        layer = GraphConvolution(fin, fout, bias=True)
        layer.weight.data = torch.from_numpy(w.copy())
        layer.bias.data = torch.from_numpy(b.copy())
        xt = torch.from_numpy(x.copy()).requires_grad_(True)
        y = layer(xt, a)
        y.backward(torch.from_numpy(dy))
        out[f"{tag}_out"] = y.detach().numpy()
        out[f"{tag}_dW"] = layer.weight.grad.numpy().copy()
        out[f"{tag}_db"] = layer.bias.grad.numpy().copy()
        out[f"{tag}_dX"] = xt.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "gcn_layers.npz"), **out)
    print("gcn_layers.npz:", {k: v.shape for k, v in out.items()})


def make_matrix_known_answers():
    # tests/utils/test_matrix.py:33-39
    mat = np.array([[0, 1, 2], [2, 2, 4], [5, 3, 5], [3, 2, 1], [5, 6, 3]], dtype=np.float32)

    def compute_pairwise(ary, func):  # tests/utils/test_matrix.py:41-47
        size = ary.shape[0]
        out = np.zeros((size, size), dtype=np.float32)
        for i, j in itertools.product(range(size), range(size)):
            if i <= j:
                out[i, j] = out[j, i] = func(ary[i], ary[j])
        return out

    ans = {
        "mat": mat.tolist(),
        "euclidean": compute_pairwise(mat, scipy.spatial.distance.euclidean).tolist(),
        "pearson": compute_pairwise(mat, lambda x, y: 1 - scipy.stats.pearsonr(x, y)[0]).tolist(),
        "spearman": compute_pairwise(mat, lambda x, y: 1 - scipy.stats.spearmanr(x, y)[0]).tolist(),
        # tests/utils/test_matrix.py:9-29
        "normalize_input": [[1, 1], [4, 4]],
        "normalize": {
            "normalize_axis0": [[0.2, 0.2], [0.8, 0.8]], "normalize_axis1": [[0.5, 0.5], [0.5, 0.5]],
            "standardize_axis0": [[-1, -1], [1, 1]], "standardize_axis1": [[0, 0], [0, 0]],
            "minmax_axis0": [[0, 0], [1, 1]], "minmax_axis1": [[0, 0], [0, 0]],
        },
    }
    with open(os.path.join(HERE, "matrix_known_answers.json"), "w") as f:
        json.dump(ans, f, indent=1)
    print("matrix_known_answers.json written")


if __name__ == "__main__":
    if not ref_extract.available():
        raise SystemExit("reference tree not found: golden vectors can only be generated in the build container")
    torch.manual_seed(0)
    make_gcn_layers()
    make_matrix_known_answers()
