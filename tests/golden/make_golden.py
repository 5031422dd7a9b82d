"""Generate the committed golden vectors from the REFERENCE's own code (run in the build container only).

    python tests/golden/make_golden.py

* gcn_layers.npz — the reference's GNNLayer (dance/modules/single_modality/clustering/scdsc.py:475-501) and
  GraphConvolution (dance/modules/spatial/spatial_domain/spagcn.py:337-366), AST-extracted from
  /root/reference and executed on torch-CPU: forward outputs and autograd gradients on seeded inputs.
* model_heads.npz — SimpleGCDEC (dance/modules/spatial/spatial_domain/spagcn.py:369-425) and ScDSCModel
  (dance/modules/single_modality/clustering/scdsc.py:339-472) forward passes of the reference's own classes.
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


def make_models():
    """model_heads.npz — the reference's own SimpleGCDEC.forward / target_distribution / loss_function
    (spagcn.py:391-425) and ScDSCModel.forward (scdsc.py:418-472, eval mode), AST-extracted and run on torch-CPU."""
    from torch.nn import Linear
    from torch.nn.parameter import Parameter
    rng = np.random.default_rng(77)
    out = {}
    # ---- SimpleGCDEC -------------------------------------------------------------------------------------
    spa = "dance/modules/spatial/spatial_domain/spagcn.py"
    GC = ref_extract.extract(spa, "GraphConvolution")
    GCDEC = ref_extract.extract(spa, "SimpleGCDEC", {"GraphConvolution": GC})
    n, d, c = 80, 12, 4
    x = rng.standard_normal((n, d)).astype(np.float32)
    adj = np.exp(-rng.uniform(0, 4, (n, n))).astype(np.float32)
    m = GCDEC(d, d)
    m.mu = Parameter(torch.from_numpy(rng.standard_normal((c, d)).astype(np.float32)))
    z, q = m.forward(torch.from_numpy(x), torch.from_numpy(adj))
    p = m.target_distribution(q)
    loss = m.loss_function(p.data, q)
    loss.backward()
    out.update(gcdec_x=x, gcdec_adj=adj, gcdec_w=m.gc.weight.detach().numpy(), gcdec_b=m.gc.bias.detach().numpy(),
               gcdec_mu=m.mu.detach().numpy(), gcdec_z=z.detach().numpy(), gcdec_q=q.detach().numpy(),
               gcdec_p=p.detach().numpy(), gcdec_loss=np.array(loss.item(), dtype=np.float32),
               gcdec_dw=m.gc.weight.grad.numpy(), gcdec_dmu=m.mu.grad.numpy())
    # ---- ScDSCModel --------------------------------------------------------------------------------------
    sc_ = "dance/modules/single_modality/clustering/scdsc.py"
    ns = {"Linear": Linear, "get_device": lambda d: "cpu"}
    for name in ("GNNLayer", "AE", "MeanAct", "DispAct"):
        ns[name] = ref_extract.extract(sc_, name, ns)
    ns["ZINBLoss"] = ref_extract.extract("dance/utils/loss.py", "ZINBLoss")
    Model = ref_extract.extract(sc_, "ScDSCModel", ns)
    torch.manual_seed(5)
    kw = dict(sigma=0.4, n_enc_1=24, n_enc_2=16, n_enc_3=16, n_dec_1=16, n_dec_2=16, n_dec_3=24, n_z1=16, n_z2=12, n_z3=8,
              n_clusters=5, n_input=20, v=1, device="cpu")
    model = Model(**kw).eval()
    n = 70
    xs = rng.standard_normal((n, 20)).astype(np.float32)
    a = sp.random(n, n, density=0.1, random_state=3, format="csr", dtype=np.float32)
    a = (a + a.T + sp.eye(n)).tocsr().astype(np.float32)
    with torch.no_grad():
        x_bar, q, predict, z3, _mean, _disp, _pi, _ = model(torch.from_numpy(xs), scipy_to_torch_coo(a))
    out.update(scdsc_x=xs, scdsc_adj_indptr=a.indptr.astype(np.int32), scdsc_adj_indices=a.indices.astype(np.int32),
               scdsc_adj_data=a.data, scdsc_x_bar=x_bar.numpy(), scdsc_q=q.numpy(), scdsc_predict=predict.numpy(),
               scdsc_z3=z3.numpy(), scdsc_mean=_mean.numpy(), scdsc_disp=_disp.numpy(), scdsc_pi=_pi.numpy())
    for k, v in model.state_dict().items():
        out["scdsc_sd::" + k] = v.numpy()
    out["scdsc_kw"] = np.array(json.dumps({k: v for k, v in kw.items() if k != "device"}))
    np.savez_compressed(os.path.join(HERE, "model_heads.npz"), **out)
    print("model_heads.npz:", len(out), "arrays")


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
    make_models()
    make_matrix_known_answers()
