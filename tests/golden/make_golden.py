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


def make_graph_builders():
    """graph_builders.npz: outputs of the reference's OWN graph-builder / message-passing code, lifted method by method
    (oracle.ref_extract.extract_method) and run on torch-CPU.  Where that code calls into dgl, oracle.ref_extract's
    DGLStubGraph stands in for the DGL storage semantics (edge ids in insertion order, mean over in-edges); sklearn is
    the installed one.  Pins oracle.graphs.cell_feature_graph / heteronet_edges, oracle.sage.* and oracle.spagcn.*."""
    import logging
    import types
    rng = np.random.default_rng(7)
    out = {}
    dgl = ref_extract.dgl_stub()
    log = logging.getLogger("reference")

    # ---- CellFeatureGraph.__call__ (cell_feature_graph.py:34-79), both normalize_edges settings ----------------
    call = ref_extract.extract_method("dance/transforms/graph/cell_feature_graph.py", "CellFeatureGraph", "__call__", {"dgl": dgl})
    n_cells, n_genes, d = 13, 8, 5
    x = ((rng.random((n_cells, n_genes)) < 0.35) * rng.integers(1, 7, (n_cells, n_genes))).astype(np.float32)
    x[3] = 0      # a cell expressing nothing
    x[:, 5] = 0   # a gene expressed nowhere
    gene_f, cell_f = rng.standard_normal((n_genes, d)).astype(np.float32), rng.standard_normal((n_cells, d)).astype(np.float32)

    def get_feature(return_type="default", mod=None, channel=None, channel_type=None):
        if channel_type == "varm":
            return torch.from_numpy(gene_f)
        if channel_type == "obsm":
            return torch.from_numpy(cell_f)
        return x
    out["cfg_x"], out["cfg_gene_feat"], out["cfg_cell_feat"] = x, gene_f, cell_f
    for norm in (False, True):
        data = types.SimpleNamespace(get_feature=get_feature, data=types.SimpleNamespace(uns={}))
        self = types.SimpleNamespace(mod=None, normalize_edges=norm, cell_feature_channel="f", gene_feature_channel="f",
                                     logger=log, out="g")
        call(self, data)
        g = data.data.uns["g"]
        src, dst = g.edges()
        tag = f"cfg_norm{int(norm)}_"
        out[tag + "src"], out[tag + "dst"] = src.numpy(), dst.numpy()
        out[tag + "weight"] = g.edata["weight"].numpy().ravel()
        out[tag + "cell_id"], out[tag + "feat_id"] = g.ndata["cell_id"].numpy(), g.ndata["feat_id"].numpy()
        out[tag + "features"] = g.ndata["features"].numpy()

    # ---- AdaptiveSAGE.message_func / forward (gnn.py:62-96) on a block with all four edge kinds ----------------
    gnn = "dance/models/nn/gnn.py"
    message_func = ref_extract.extract_method(gnn, "AdaptiveSAGE", "message_func")
    forward = ref_extract.extract_method(gnn, "AdaptiveSAGE", "forward", {"dgl": dgl})
    G, n_src, n_dst, dim, dim_out = 6, 14, 9, 7, 4
    cid = np.concatenate((np.arange(G), -np.ones(n_src - G))).astype(np.int32)
    perm = rng.permutation(n_src)
    cid = cid[perm]                                     # genes and cells interleaved among the source nodes
    e = 60
    e_src, e_dst = rng.integers(0, n_src, e), rng.integers(0, n_dst, e)
    e_dst[e_dst == 4] = 3                               # destination 4 stays isolated (mean -> 0)
    w = (rng.random(e).astype(np.float32) + 0.1)[:, None]
    alpha = (rng.random((G + 2, 1)).astype(np.float32) + 0.5)
    h = rng.standard_normal((n_src, dim)).astype(np.float32)
    lin = torch.nn.Linear(dim, dim_out)
    self = types.SimpleNamespace(gene_num=G, alpha=torch.from_numpy(alpha), layers=[torch.nn.Identity(), lin, torch.nn.ReLU(), torch.nn.Identity()])
    self.message_func = lambda edges: message_func(self, edges)
    blk = ref_extract.DGLStubGraph(e_src, e_dst, num_src=n_src, num_dst=n_dst)
    blk.srcdata["cell_id"], blk.dstdata["cell_id"] = torch.from_numpy(cid), torch.from_numpy(cid[:n_dst])
    blk.edata["weight"] = torch.from_numpy(w)
    with torch.no_grad():
        z = forward(self, blk, torch.from_numpy(h))
    edges = types.SimpleNamespace(src={"h": torch.from_numpy(h)[e_src], "cell_id": torch.from_numpy(cid)[e_src]},
                                  dst={"cell_id": torch.from_numpy(cid[:n_dst])[e_dst]}, data={"weight": torch.from_numpy(w)})
    out.update(sage_src=e_src, sage_dst=e_dst, sage_w=w.ravel(), sage_cid=cid, sage_alpha=alpha, sage_h=h, sage_n_dst=np.int64(n_dst),
               sage_m=message_func(self, edges)["m"].numpy(), sage_neigh=blk.dstdata["neigh"].numpy(), sage_z=z.numpy(),
               sage_lin_w=lin.weight.detach().numpy(), sage_lin_b=lin.bias.detach().numpy())

    # ---- WeightedGraphConv.forward (graphsc.py:428-484; GraphConv defaults norm="both", bias) on the same block --------
    gsc = "dance/modules/single_modality/clustering/graphsc.py"
    esel = ref_extract.extract_method(gsc, "WeightedGraphConv", "edge_selection_simple")
    wfwd = ref_extract.extract_method(gsc, "WeightedGraphConv", "forward",
                                      {"fn": dgl.function, "expand_as_pair": lambda feat, g: (feat, feat[:g.number_of_dst_nodes()]),
                                       "DGLError": RuntimeError})
    e_dst2 = e_dst.copy()
    e_dst2[:n_dst] = np.arange(n_dst)                   # every destination has an in-edge (allow_zero_in_degree=False)
    wmat = (rng.standard_normal((dim, dim_out)) * 0.3).astype(np.float32)
    bvec = rng.standard_normal(dim_out).astype(np.float32)
    out.update(wgc_src=e_src, wgc_dst=e_dst2, wgc_w=w.ravel(), wgc_feat=h, wgc_weight=wmat, wgc_bias=bvec)
    for agg in ("sum", "mean"):
        blk2 = ref_extract.DGLStubGraph(e_src, e_dst2, num_src=n_src, num_dst=n_dst)
        blk2.edata["weight"] = torch.from_numpy(w)
        conv = types.SimpleNamespace(_allow_zero_in_degree=False, _norm="both", weight=torch.from_numpy(wmat),
                                     bias=torch.from_numpy(bvec), _activation=torch.relu)
        conv.edge_selection_simple = lambda edges: esel(conv, edges)
        out["wgc_out_" + agg] = wfwd(conv, blk2, torch.from_numpy(h), agg=agg).numpy()

    # ---- HeteronetGraph.build_graph (heteronet_graph.py:27-40) ------------------------------------------------
    from sklearn.neighbors import NearestNeighbors
    build = ref_extract.extract_method("dance/transforms/graph/heteronet_graph.py", "HeteronetGraph", "build_graph",
                                       {"NearestNeighbors": NearestNeighbors})
    feats = rng.standard_normal((40, 6)).astype(np.float32)
    out["het_feats"], out["het_edges"] = feats, build(None, feats, knears=5)

    # ---- calculate_p / search_l (spagcn.py:249-287) ------------------------------------------------------------
    spa = "dance/modules/spatial/spatial_domain/spagcn.py"
    calculate_p = ref_extract.extract(spa, "calculate_p")
    search_l = ref_extract.extract(spa, "search_l", {"calculate_p": calculate_p, "logger": log})
    xy = rng.random((60, 2)).astype(np.float32) * 20
    adj = np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(-1)).astype(np.float32)
    out["spa_adj"] = adj
    out["spa_p_at"] = np.array([[l, calculate_p(adj, l)] for l in (0.3, 1.0, 2.5, 40.0)], dtype=np.float64)
    out["spa_search"] = np.array([[p, (search_l(p, adj) or np.nan)] for p in (0.5, 2.0, 1e-9, 1e4)], dtype=np.float64)
    # ---- SpaGCNGraph.__call__ (spatial_graph.py:36-62): the [x, y, z] coordinates handed to pairwise_distance -----------
    sg = "dance/transforms/graph/spatial_graph.py"
    captured = {}

    def capture_pairwise(xyz, dist_func_id=0):
        captured["xyz"], captured["dist_func_id"] = xyz, dist_func_id
        return None
    spa_call = ref_extract.extract_method(sg, "SpaGCNGraph", "__call__", {"pairwise_distance": capture_pairwise})
    n_spots = 30
    sxy = rng.integers(0, 50, (n_spots, 2)).astype(np.float32)
    spix = rng.integers(0, 64, (n_spots, 2)).astype(np.int64)
    spix[0], spix[1] = (0, 1), (63, 62)  # windows clipped by the image border
    img = rng.integers(0, 256, (64, 64, 3)).astype(np.uint8)
    feats_by_channel = {"spatial": sxy, "spatial_pixel": spix, "image": img}
    sdata = types.SimpleNamespace(get_feature=lambda return_type, channel, channel_type: feats_by_channel[channel],
                                  data=types.SimpleNamespace(obsp={}))
    sself = types.SimpleNamespace(alpha=1.5, beta=9, channels=("spatial", "spatial_pixel", "image"),
                                  channel_types=("obsm", "obsm", "uns"), logger=log, out="SpaGCNGraph")
    spa_call(sself, sdata)
    out.update(spg_xy=sxy, spg_xy_pixel=spix, spg_img=img, spg_alpha=np.float64(1.5), spg_beta=np.int64(9), spg_xyz=captured["xyz"],
               spg_dist_func_id=np.int64(captured["dist_func_id"]))

    # ---- StagateGraph.__call__ (spatial_graph.py:143-151), both models ------------------------------------------------------
    st_call = ref_extract.extract_method(sg, "StagateGraph", "__call__", {"NearestNeighbors": NearestNeighbors})
    pts = (rng.random((50, 2)) * 10).astype(np.float32)
    out["stg_xy"] = pts
    for model, kw in (("radius", dict(radius=1.7)), ("knn", dict(n_neighbors=4))):
        tdata = types.SimpleNamespace(get_feature=lambda return_type, channel, channel_type: pts, data=types.SimpleNamespace(obsp={}))
        tself = types.SimpleNamespace(model_name=model, radius=kw.get("radius", 1), n_neighbors=kw.get("n_neighbors", 5),
                                      channel="spatial_pixel", channel_type="obsm", out="StagateGraph")
        st_call(tself, tdata)
        out["stg_" + model] = np.asarray(tdata.data.obsp["StagateGraph"].todense(), dtype=np.float32)

    # ---- refine (spagcn.py:290-334) ------------------------------------------------------------------------------------------
    refine = ref_extract.extract(spa, "refine", {"pd": __import__("pandas"), "logger": log})
    hexxy = np.array([[i + 0.5 * (j % 2), j * 0.866] for i in range(7) for j in range(7)], dtype=np.float32)
    dis = np.sqrt(((hexxy[:, None] - hexxy[None]) ** 2).sum(-1)).astype(np.float32)
    pred0 = (hexxy[:, 0] > 3).astype(np.int64)
    pred0[[8, 24, 40]] = 1 - pred0[[8, 24, 40]]  # isolated mislabelled spots get voted back
    ids = [f"s{i}" for i in range(len(pred0))]
    out.update(ref_dis=dis, ref_pred=pred0, ref_refined_hexagon=np.asarray(refine(ids, pred0, dis, shape="hexagon"), dtype=np.int64),
               ref_refined_square=np.asarray(refine(ids, pred0, dis, shape="square"), dtype=np.int64))

    np.savez_compressed(os.path.join(HERE, "graph_builders.npz"), **out)
    print("graph_builders.npz:", len(out), "arrays")


def make_graphsc():
    """graphsc.npz — the reference's own GCNAE / InnerProductDecoder / WeightedGraphConv classes and the GraphSC.fit loop
    (graphsc.py:148-246,274-484), AST-lifted and run on torch-CPU over the DGL stub (blocks of the full-neighbour
    sampler, ``adjacency_matrix``; dgl.nn.GraphConv's constructor restated from the DGL 1.1.3 documentation).  Dropout
    is switched off (GCNAE dropout=0, decoder.dropout=0) so the run is deterministic; seed order = randperm of a seeded
    generator that the GPU test hands to the product's loader."""
    import logging
    import types

    import torch.nn as nn
    import torch.nn.functional as F
    rng = np.random.default_rng(11)
    gsc = "dance/modules/single_modality/clustering/graphsc.py"
    out = {}

    class GraphConv(nn.Module):  # dgl.nn.pytorch.GraphConv.__init__ / reset_parameters
        def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None, allow_zero_in_degree=False):
            super().__init__()
            self._in_feats, self._out_feats, self._norm, self._allow_zero_in_degree = in_feats, out_feats, norm, allow_zero_in_degree
            self.weight = nn.Parameter(torch.Tensor(in_feats, out_feats)) if weight else None
            self.bias = nn.Parameter(torch.Tensor(out_feats)) if bias else None
            nn.init.xavier_uniform_(self.weight)
            nn.init.zeros_(self.bias)
            self._activation = activation

    n_cells, n_genes, d = 40, 12, 6
    x = ((rng.random((n_cells, n_genes)) < 0.3) * rng.uniform(0.2, 2.0, (n_cells, n_genes))).astype(np.float32)
    x[np.arange(n_cells), rng.integers(0, n_genes, n_cells)] = 1.0   # every cell expresses something
    gene_f, cell_f = rng.standard_normal((n_genes, d)).astype(np.float32), rng.standard_normal((n_cells, d)).astype(np.float32)
    out.update(gsc_x=x, gsc_gene_feat=gene_f, gsc_cell_feat=cell_f)

    def build_graph(dgl):
        call = ref_extract.extract_method("dance/transforms/graph/cell_feature_graph.py", "CellFeatureGraph", "__call__", {"dgl": dgl})
        get_feature = lambda return_type="default", mod=None, channel=None, channel_type=None: (
            torch.from_numpy(gene_f) if channel_type == "varm" else torch.from_numpy(cell_f) if channel_type == "obsm" else x)
        data = types.SimpleNamespace(get_feature=get_feature, data=types.SimpleNamespace(uns={}))
        call(types.SimpleNamespace(mod=None, normalize_edges=False, cell_feature_channel="f", gene_feature_channel="f",
                                   logger=logging.getLogger("reference"), out="g"), data)
        return data.data.uns["g"]

    kw = dict(agg="sum", activation="relu", in_feats=d, n_hidden=1, hidden_dim=10, hidden_1=5, hidden_2=0, dropout=0.0,
              n_layers=1, hidden_relu=False, hidden_bn=False)
    out["gsc_kw"] = np.array(json.dumps(kw))
    for tag, batch_size, agg in (("full", 64, "sum"), ("mb", 16, "sum"), ("mean", 16, "mean")):
        gen = torch.Generator().manual_seed(123)
        dgl = ref_extract.dgl_stub(shuffle_generator=gen)
        ns = {"GraphConv": GraphConv, "fn": dgl.function, "DGLError": RuntimeError,
              "expand_as_pair": lambda feat, g: (feat, feat[:g.number_of_dst_nodes()])}
        WGC = ref_extract.extract(gsc, "WeightedGraphConv", ns)
        IPD = ref_extract.extract(gsc, "InnerProductDecoder")
        GCNAE = ref_extract.extract(gsc, "GCNAE", {"WeightedGraphConv": WGC, "InnerProductDecoder": IPD})
        from typing import Any, Optional
        fit = ref_extract.extract_method(gsc, "GraphSC", "fit", {"dgl": dgl, "get_device": lambda dev: "cpu", "tqdm": lambda it: it,
                                                                 "BCELoss": F.binary_cross_entropy_with_logits, "Any": Any, "Optional": Optional})
        torch.manual_seed(3)
        model = GCNAE(**{**kw, "agg": agg})
        model.decoder.dropout = 0.0
        with torch.no_grad():
            model.layer1.bias.uniform_(-0.1, 0.1)
        for k, v in model.state_dict().items():
            out[f"gsc_{tag}_sd0::{k}"] = v.numpy().copy()
        g = build_graph(dgl)
        # one forward on the block of ALL cells before training: pins GCNAE.forward / InnerProductDecoder
        _, _, blocks = ref_extract.stub_full_in_block(g, torch.arange(n_genes, n_genes + n_cells))
        with torch.no_grad():
            logits, emb = model.forward(blocks, blocks[0].srcdata["features"])
        out[f"gsc_{tag}_logits0"], out[f"gsc_{tag}_emb0"] = logits.numpy(), emb.numpy()
        # the reference's training loop.  Its ``losses`` list is local, so record the tensor ``backward`` is called on.
        import unittest.mock as mock
        losses = []
        self = types.SimpleNamespace(model=model, n_layers=1, num_workers=0, device="cpu", score=None)
        real_backward = torch.Tensor.backward

        def recording_backward(t, *a, **k):
            losses.append(float(t.detach()))
            return real_backward(t, *a, **k)
        with mock.patch.object(torch.Tensor, "backward", recording_backward):
            fit(self, g, epochs=3, lr=1e-2, batch_size=batch_size)
        out[f"gsc_{tag}_losses"] = np.array(losses, dtype=np.float64)   # norm * BCE per batch (graphsc.py:211-216)
        out[f"gsc_{tag}_z"] = np.asarray(self.z, dtype=np.float32)
        for k, v in model.state_dict().items():
            out[f"gsc_{tag}_sd1::{k}"] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "graphsc.npz"), **out)
    print("graphsc.npz:", len(out), "arrays")


def scdeepsort_reference(dgl, save_dir, log):
    """The reference's own ScDeepSort (scdeepsort.py:91-349) as a class assembled from its AST-lifted methods (the real class
    cannot be defined here: its base needs the dance registry), with its GNN (:26-88) and AdaptiveSAGE (gnn.py:8-96) lifted whole.
    ``log`` receives every value ``cal_loss`` / ``evaluate`` return and the freshly initialised ``state_dict``.  Also used by
    tests/test_oracle_layers.py to re-derive the golden file."""
    import logging
    from contextlib import nullcontext
    from copy import deepcopy
    from pathlib import Path
    from typing import Optional

    import torch.nn as nn
    sds = "dance/modules/single_modality/cell_type_annotation/scdeepsort.py"
    SAGE = ref_extract.extract("dance/models/nn/gnn.py", "AdaptiveSAGE", {"dgl": dgl, "logger": logging.getLogger("reference")})
    GNN = ref_extract.extract(sds, "GNN", {"AdaptiveSAGE": SAGE, "Optional": Optional})

    def recording_gnn(*a, **k):
        m = GNN(*a, **k)
        log.setdefault("sd0", {k2: v.detach().clone() for k2, v in m.state_dict().items()})
        return m
    ns = {"dgl": dgl, "DataLoader": dgl.dataloading.DataLoader, "NeighborSampler": dgl.dataloading.NeighborSampler, "GNN": recording_gnn,
          "deepcopy": deepcopy, "nullcontext": nullcontext, "print": lambda *a, **k: None}
    methods = {m: ref_extract.extract_method(sds, "ScDeepSort", m, ns)
               for m in ("fit", "cal_loss", "evaluate", "save_model", "predict_proba", "predict")}

    def recorded(name):
        fn = methods[name]

        def wrapper(self, *a, **k):
            r = fn(self, *a, **k)
            log.setdefault(name, []).append(r)
            return r
        return wrapper
    methods["cal_loss"], methods["evaluate"] = recorded("cal_loss"), recorded("evaluate")
    Ref = type("ReferenceScDeepSort", (), methods)

    def build(dim_in, dim_hid, batch_size):  # the attributes ScDeepSort.__init__ sets (:115-133), without its mkdir in the CWD
        m = Ref()
        m.dense_dim, m.hidden_dim, m.n_layers, m.dropout = dim_in, dim_hid, 1, 0
        m.species, m.tissue, m.batch_size, m.device = "golden", "scds", batch_size, "cpu"
        m.prj_path, m.save_path = Path(save_dir), Path(save_dir)
        return m
    return build


def scdeepsort_inputs():
    """Seeded inputs of scdeepsort.npz: 300 cells of 4 expression programmes over 40 genes, 16-d gene features, cell features =
    row-normalised expression @ gene features (what WeightedFeaturePCA hands PCACellFeatureGraph, cell_feature.py:66-67)."""
    rng = np.random.default_rng(21)
    n_cells, n_genes, d, n_types = 300, 40, 16, 4
    types = rng.integers(0, n_types, n_cells)
    rates = rng.gamma(0.5, 1.0, (n_types, n_genes)) * 1.5
    x = rng.poisson(rates[types]).astype(np.float32)
    x[np.arange(n_cells), rng.integers(0, n_genes, n_cells)] += 1.0   # every cell expresses something
    gene_f = rng.standard_normal((n_genes, d)).astype(np.float32)
    cell_f = ((x / x.sum(1, keepdims=True)) @ gene_f).astype(np.float32)
    return x, gene_f, cell_f, types.astype(np.int64)


def scdeepsort_graph(dgl, x, gene_f, cell_f):
    """CellFeatureGraph.__call__ of the reference (normalize_edges=True, the scDeepSort pipeline's default) on the stub."""
    import logging
    import types as pytypes
    call = ref_extract.extract_method("dance/transforms/graph/cell_feature_graph.py", "CellFeatureGraph", "__call__", {"dgl": dgl})
    get_feature = lambda return_type="default", mod=None, channel=None, channel_type=None: (
        torch.from_numpy(gene_f) if channel_type == "varm" else torch.from_numpy(cell_f) if channel_type == "obsm" else x)
    data = pytypes.SimpleNamespace(get_feature=get_feature, data=pytypes.SimpleNamespace(uns={}))
    call(pytypes.SimpleNamespace(mod=None, normalize_edges=True, cell_feature_channel="f", gene_feature_channel="f",
                                 logger=logging.getLogger("reference"), out="g"), data)
    return data.data.uns["g"]


def scdeepsort_run(batch_size, epochs=3, lr=1e-2, hid=12, seed_model=5, seed_order=77):
    """One run of the reference's ScDeepSort.fit + predict_proba + predict; returns the arrays of one scdeepsort.npz case.
    Every random permutation — the train / validation split (scdeepsort.py:157, ``torch.randperm`` on the global RNG) and the
    stub loader's per-epoch seed orders — is drawn from ONE seeded generator in program order, which is the order the product's
    ``shuffle_generator`` is consumed in (split, then per epoch: training loader, evaluate(train) loader, evaluate(val) loader)."""
    import tempfile
    import unittest.mock as mock
    x, gene_f, cell_f, labels = scdeepsort_inputs()
    gen = torch.Generator().manual_seed(seed_order)
    dgl = ref_extract.dgl_stub(shuffle_generator=gen)
    g = scdeepsort_graph(dgl, x, gene_f, cell_f)
    log = {}
    real_randperm = torch.randperm
    with tempfile.TemporaryDirectory() as tmp:
        m = scdeepsort_reference(dgl, tmp, log)(gene_f.shape[1], hid, batch_size)
        torch.manual_seed(seed_model)
        with mock.patch.object(torch, "randperm", lambda n, *a, **k: real_randperm(int(n), *a, **({"generator": gen, **k}))):
            m.fit(g, torch.from_numpy(labels), epochs=epochs, lr=lr, weight_decay=0, val_ratio=0.2)
            prob = m.predict_proba(g)
            pred, unsure = m.predict(g, unsure_rate=2.0, return_unsure=True)
            pred_loose, unsure_loose = m.predict(g, unsure_rate=1.4, return_unsure=True)
    ev = np.array([[c, u, a] for c, u, a in log["evaluate"]], dtype=np.float64)   # per epoch: evaluate(train), evaluate(val)
    out = {"losses": np.array(log["cal_loss"], dtype=np.float64), "eval": ev, "prob": prob.astype(np.float32), "pred": pred.astype(np.int64),
           "unsure": unsure.astype(np.bool_), "pred_loose": pred_loose.astype(np.int64), "unsure_loose": unsure_loose.astype(np.bool_)}
    for k, v in log["sd0"].items():
        out[f"sd0::{k}"] = v.numpy().copy()
    for k, v in m.model.state_dict().items():          # after fit: the best-validation checkpoint (scdeepsort.py:204-206)
        out[f"sd1::{k}"] = v.numpy().copy()
    return out


def make_scdeepsort():
    """scdeepsort.npz — the reference's own ScDeepSort.fit / cal_loss / evaluate / predict_proba / predict (scdeepsort.py:142-349)
    with its GNN and AdaptiveSAGE, AST-lifted and run on torch-CPU over the DGL stub: per-epoch summed-CE loss, (correct, unsure,
    accuracy) of evaluate on the training and validation cells (the raw-logit "unsure" rule of :280-281), the best-validation
    state_dict fit ends with, predict_proba and predict(..., return_unsure=True).  Cases: "one" = every epoch is one batch
    (batch_size >= cells: the seed order only permutes a sum), "mb" = batches of 64 in the seeded loader order, "peak" = the
    same at lr 0.1, where validation accuracy peaks at the middle epoch (so restoring the best checkpoint is observable)."""
    x, gene_f, cell_f, labels = scdeepsort_inputs()
    cases = {"one": {"batch_size": 512, "epochs": 3, "lr": 1e-2}, "mb": {"batch_size": 64, "epochs": 3, "lr": 1e-2},
             "peak": {"batch_size": 64, "epochs": 3, "lr": 0.1}}
    out = {"x": x, "gene_feat": gene_f, "cell_feat": cell_f, "labels": labels,
           "kw": np.array(json.dumps({"hid": 12, "seed_model": 5, "seed_order": 77, "cases": cases}))}
    for tag, c in cases.items():
        for k, v in scdeepsort_run(c["batch_size"], epochs=c["epochs"], lr=c["lr"]).items():
            out[f"{tag}_{k}"] = v
    val_acc = out["peak_eval"][1::2, 2]
    assert val_acc.argmax() < len(val_acc) - 1, val_acc   # "peak": the best validation epoch is NOT the last one
    np.savez_compressed(os.path.join(HERE, "scdeepsort.npz"), **out)
    print("scdeepsort.npz:", len(out), "arrays;", {t: (out[f"{t}_losses"].round(4).tolist(), out[f"{t}_eval"][:, 2].round(4).tolist()) for t in cases})


def make_scheteronet():
    """scheteronet.npz — the reference's own HeteroNet / scHeteroNet classes (scheteronet.py:281-789), AST-lifted and run on
    torch-CPU over stand-ins for torch_sparse / torch_geometric (oracle.ref_extract.pyg_stub): init_adj's two normalised
    adjacencies, the eval-mode forward with the ZINB decoder, energy propagation (one- and two-hop), and one ``fit`` step
    (dropout = 0, ZINB loss on, contrastive term off — it draws a random mask)."""
    import types

    import scipy.sparse
    import torch.nn as nn
    import torch.nn.functional as F
    from typing import Any, Mapping, Optional, Tuple, Union
    sh = "dance/modules/single_modality/cell_type_annotation/scheteronet.py"
    pyg = ref_extract.pyg_stub()
    ns = {"SparseTensor": pyg.SparseTensor, "matmul": pyg.matmul, "gcn_norm": pyg.gcn_norm, "JumpingKnowledge": pyg.JumpingKnowledge,
          "degree": pyg.degree, "scipy": scipy, "BaseClassificationMethod": object, "LogLevel": str, "Any": Any, "Mapping": Mapping,
          "Optional": Optional, "Tuple": Tuple, "Union": Union, "logger": __import__("logging").getLogger("reference")}
    for name in ("contrastive_loss", "ZINBLoss", "MLP", "HetConv", "MeanAct", "DispAct", "ZINBDecoder", "HeteroNet", "scHeteroNet"):
        ns[name] = ref_extract.extract(sh, name, ns)
    rng = np.random.default_rng(5)
    n, d, c, hid, k = 70, 18, 4, 8, 4
    lab = rng.integers(0, c, n)
    x = (rng.standard_normal((n, d)) + np.eye(c, d)[lab] * 2).astype(np.float32)
    from sklearn.neighbors import NearestNeighbors
    idx = NearestNeighbors(n_neighbors=k + 1).fit(x).kneighbors(x, return_distance=False)   # HeteronetGraph: self is neighbour 0
    edge_index = np.stack((np.repeat(np.arange(n), k + 1), idx.reshape(-1))).astype(np.int64)
    out = dict(sh_x=x, sh_edge_index=edge_index, sh_y=lab.astype(np.int64), sh_dims=np.array([n, d, c, hid], dtype=np.int64))
    torch.manual_seed(11)
    model = ns["scHeteroNet"](d, c, torch.from_numpy(edge_index), n, hid, 2, 0.0, True, "cpu", 100.0)
    enc = model.encoder
    out["sh_adj_t"] = np.asarray(enc.adj_t.csr.todense(), dtype=np.float32)
    out["sh_adj_t2"] = np.asarray(enc.adj_t2.csr.todense(), dtype=np.float32)
    for kk, v in model.state_dict().items():
        out["sh_sd0::" + kk] = v.numpy().copy()
    ds = types.SimpleNamespace(x=torch.from_numpy(x), edge_index=torch.from_numpy(edge_index), y=torch.from_numpy(lab)[:, None],
                               splits={"train": torch.arange(0, n, 2)}, node_idx=torch.arange(n))
    model.eval()
    with torch.no_grad():
        h, m_, d_, p_ = enc(ds.x, ds.edge_index, decoder=True)
        e = torch.from_numpy(rng.standard_normal(n).astype(np.float32))
        out.update(sh_logits=h.numpy(), sh_mean=m_.numpy(), sh_disp=d_.numpy(), sh_pi=p_.numpy(), sh_e=e.numpy(),
                   sh_prop=model.propagation(e, ds.edge_index, 2, 0.5).numpy(),
                   sh_prop2=model.two_hop_propagation(e, ds.edge_index, 1, 0.3).numpy(),
                   sh_detect=model.detect(ds, ds.node_idx, "cpu", 1.0, True, False, 2, 0.5).numpy())
    # one training step (train mode: BatchNorm on batch statistics; dropout 0)
    counts = rng.poisson(np.exp(x[:, :d] * 0.3)).astype(np.float32)
    adata = types.SimpleNamespace(raw=types.SimpleNamespace(X=counts), obs=types.SimpleNamespace(size_factors=(counts.sum(1) / counts.sum(1).mean()).astype(np.float32)))
    out.update(sh_counts=counts, sh_size_factors=adata.obs.size_factors)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    loss = model.fit(ds, ds, True, adata, 0.5, 0.0, 0.4, nn.NLLLoss(), opt)
    out["sh_loss"] = np.array(float(loss.detach()), dtype=np.float64)
    for kk, v in model.state_dict().items():
        out["sh_sd1::" + kk] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "scheteronet.npz"), **out)
    print("scheteronet.npz:", len(out), "arrays; loss", out["sh_loss"])


def make_scdsc_fit():
    """scdsc_fit.npz — the reference's own ``ScDSC`` method class (scdsc.py:33-336: AE pre-training, the joint training loop
    with BCE + KL + MSE + ZINB, best-ARI checkpoint selection) AST-lifted together with its ScDSCModel / AE / GNNLayer /
    ZINBLoss and ``sparse_mx_to_torch_sparse_tensor`` (preprocess.py:526-532) and run on torch-CPU.  The pre-train / fix-module
    mixins (dance/modules/base.py:72-153, pure Python plumbing) are dance_amd's mirrors of them."""
    import logging
    from typing import Any, Optional, Tuple

    import pandas as pd
    from torch.nn import Linear
    from torch.optim import Adam
    from torch.utils.data import DataLoader, TensorDataset

    from dance_amd.modules.base import BaseClusteringMethod, TorchNNPretrain
    sc_ = "dance/modules/single_modality/clustering/scdsc.py"
    ns = {"Linear": Linear, "get_device": lambda d: "cpu", "Adam": Adam, "DataLoader": DataLoader, "TensorDataset": TensorDataset,
          "TorchNNPretrain": TorchNNPretrain, "BaseClusteringMethod": BaseClusteringMethod, "logger": logging.getLogger("reference"),
          "LogLevel": str, "Tuple": Tuple, "Optional": Optional, "Any": Any, "sp": sp, "pd": pd}
    for name in ("GNNLayer", "AE", "MeanAct", "DispAct"):
        ns[name] = ref_extract.extract(sc_, name, ns)
    ns["ZINBLoss"] = ref_extract.extract("dance/utils/loss.py", "ZINBLoss")
    ns["ScDSCModel"] = ref_extract.extract(sc_, "ScDSCModel", ns)
    ns["sparse_mx_to_torch_sparse_tensor"] = ref_extract.extract("dance/transforms/preprocess.py", "sparse_mx_to_torch_sparse_tensor", {"sp": sp})
    ScDSC = ref_extract.extract(sc_, "ScDSC", ns)
    rng = np.random.default_rng(21)
    n, g, c = 120, 30, 3
    lab = rng.integers(0, c, n)
    x = (rng.standard_normal((n, g)) * 0.6 + np.eye(c, g)[lab] * 2.5).astype(np.float32)
    counts = rng.poisson(np.exp(np.clip(x, -2, 2) * 0.5)).astype(np.float32)
    n_counts = counts.sum(1) + 1.0
    from sklearn.neighbors import kneighbors_graph
    a = kneighbors_graph(x, 6, include_self=True)
    a = ((a + a.T) > 0).astype(np.float32)
    adj = sp.csr_matrix(sp.diags(1.0 / np.asarray(a.sum(1)).ravel()) @ a, dtype=np.float32)   # row-normalised kNN graph
    kw = dict(sigma=0.5, n_enc_1=24, n_enc_2=16, n_enc_3=16, n_dec_1=16, n_dec_2=16, n_dec_3=24, n_z1=16, n_z2=12, n_z3=8, n_clusters=c,
              n_input=g, v=1)
    torch.manual_seed(9)
    m = ScDSC(pretrain_path=None, device="cpu", **kw)
    out = dict(sf_x=x, sf_counts=counts, sf_n_counts=n_counts.astype(np.float32), sf_y=lab.astype(np.int64), sf_adj_indptr=adj.indptr.astype(np.int32),
               sf_adj_indices=adj.indices.astype(np.int32), sf_adj_data=adj.data, sf_kw=np.array(json.dumps(kw)))
    for k, v in m.model.state_dict().items():
        out["sf_sd0::" + k] = v.numpy().copy()
    torch.manual_seed(10)
    m.fit((adj, x, counts, pd.Series(n_counts)), lab, lr=1e-3, epochs=12, pt_epochs=3, pt_batch_size=32, pt_lr=1e-3)
    out.update(sf_q=m.predict_proba(), sf_pred=m.predict().astype(np.int64))
    for k, v in m.model.state_dict().items():
        out["sf_sd1::" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "scdsc_fit.npz"), **out)
    print("scdsc_fit.npz:", len(out), "arrays; ARI-selected q", out["sf_q"].shape)


def make_sctag():
    """sctag.npz — the reference's own ScTAG / DecoderAdj / DecoderX (sctag.py:32-528) and ZINBLoss / dist_loss
    (utils/loss.py), AST-lifted and run on torch-CPU over the DGL graph stub and the restated dgl.nn.TAGConv
    (oracle.ref_extract.dgl_tagconv_stub): a forward before training, then ``fit`` with 3 pre-training and 4 training epochs
    (decoder dropout 0: the only random draw left is KMeans', seeded through numpy)."""
    import logging
    from typing import Any, Optional, Tuple

    import torch.optim as optim
    from sklearn.cluster import KMeans
    from torch.nn import Parameter

    from dance_amd.modules.base import BaseClusteringMethod, TorchNNPretrain
    st = "dance/modules/single_modality/clustering/sctag.py"
    dgl = ref_extract.dgl_stub()
    ns = {"dgl": dgl, "TAGConv": ref_extract.dgl_tagconv_stub(), "KMeans": KMeans, "Parameter": Parameter, "optim": optim,
          "get_device": lambda d: "cpu", "TorchNNPretrain": TorchNNPretrain, "BaseClusteringMethod": BaseClusteringMethod,
          "logger": logging.getLogger("reference"), "LogLevel": str, "Tuple": Tuple, "Optional": Optional, "Any": Any}
    ns["ZINBLoss"] = ref_extract.extract("dance/utils/loss.py", "ZINBLoss")
    ns["cdisttf"] = ref_extract.extract("dance/utils/loss.py", "cdisttf")
    ns["dist_loss"] = ref_extract.extract("dance/utils/loss.py", "dist_loss", ns)
    for name in ("MeanAct", "DispAct", "DecoderAdj", "DecoderX", "ScTAG"):
        ns[name] = ref_extract.extract(st, name, ns)
    rng = np.random.default_rng(31)
    n, g, c = 60, 20, 3
    lab = rng.integers(0, c, n)
    x = (rng.standard_normal((n, g)) * 0.6 + np.eye(c, g)[lab] * 2.0).astype(np.float32)
    counts = rng.poisson(np.exp(np.clip(x, -2, 2) * 0.5)).astype(np.float32)
    n_counts = counts.sum(1) + 1.0
    from sklearn.neighbors import kneighbors_graph
    a = kneighbors_graph(x, 5, include_self=False)
    w = rng.uniform(0.2, 1.0, a.nnz).astype(np.float32)     # weighted, symmetric (UMAP-connectivity-like)
    a = sp.csr_matrix((w, a.indices, a.indptr), shape=a.shape)
    adj = np.asarray(a.maximum(a.T).todense(), dtype=np.float32)
    out = dict(tg_x=x, tg_counts=counts, tg_n_counts=n_counts.astype(np.float32), tg_y=lab.astype(np.int64), tg_adj=adj)
    torch.manual_seed(4)
    m = ns["ScTAG"](n_clusters=c, k=3, hidden_dim=16, latent_dim=6, dec_dim=[12, 16, 20], dropout=0.0, device="cpu")
    m.init_model(adj, x)
    with torch.no_grad():
        m.mu.copy_(torch.from_numpy(rng.standard_normal((c, 6)).astype(np.float32)))
    for k, v in m.state_dict().items():
        out["tg_sd0::" + k] = v.numpy().copy()
    with torch.no_grad():
        xt = torch.from_numpy(x)
        adj_out, z, q, mean, disp, pi = m.forward(m.g_n, xt)
        out.update(tg_adj_out=adj_out.numpy(), tg_z=z.numpy(), tg_q=q.numpy(), tg_mean=mean.numpy(), tg_disp=disp.numpy(), tg_pi=pi.numpy(),
                   tg_enc_unweighted=m.encoder1(m.g_n, xt).numpy())
    # fit() re-initialises the model: same torch seed on both sides -> same initial weights; KMeans draws from numpy's generator
    torch.manual_seed(5)
    np.random.seed(0)
    m2 = ns["ScTAG"](n_clusters=c, k=3, hidden_dim=16, latent_dim=6, dec_dim=[12, 16, 20], dropout=0.0, device="cpu")
    m2.fit((adj, x, counts, n_counts), lab, epochs=4, pretrain_epochs=3, lr=5e-3, w_d=0.1)
    out.update(tg_fit_q=m2.predict_proba(), tg_fit_pred=m2.predict().astype(np.int64))
    for k, v in m2.state_dict().items():
        out["tg_sd1::" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "sctag.npz"), **out)
    print("sctag.npz:", len(out), "arrays")


def make_stagate():
    """stagate.npz — the reference's own GATConv / Stagate (stagate.py:31-330), AST-lifted and run on torch-CPU over the
    restated torch_geometric MessagePassing / softmax (oracle.ref_extract.pyg_message_passing_stub): a GATConv forward +
    backward (output, attention weights, gradients of x, lin_src, att_src, att_dst), the auto-encoder forward, and
    ``pretrain`` for 5 epochs (deterministic: no dropout)."""
    import logging
    from typing import Any, Optional, Tuple

    from torch import Tensor
    from torch.nn import Parameter

    from dance_amd.modules.base import BaseClusteringMethod, BasePretrain
    sg = "dance/modules/spatial/spatial_domain/stagate.py"
    pyg = ref_extract.pyg_message_passing_stub()
    ns = {"MessagePassing": pyg.MessagePassing, "softmax": pyg.softmax, "add_self_loops": pyg.add_self_loops,
          "remove_self_loops": pyg.remove_self_loops, "Tensor": Tensor, "Parameter": Parameter, "get_device": lambda d: "cpu",
          "BasePretrain": BasePretrain, "BaseClusteringMethod": BaseClusteringMethod, "logger": logging.getLogger("reference"), "LogLevel": str,
          "Tuple": Tuple, "Optional": Optional, "Any": Any, "tqdm": lambda it: it}
    ns["GATConv"] = ref_extract.extract(sg, "GATConv", ns)
    Stagate = ref_extract.extract(sg, "Stagate", ns)
    rng = np.random.default_rng(41)
    n, d = 80, 24
    xy = rng.random((n, 2)) * 10
    dist = np.sqrt(((xy[:, None] - xy[None])**2).sum(-1))
    a = (dist < 1.9)                                   # StagateGraph(radius): self included
    src, dst = np.nonzero(a)
    edge_index = np.stack((src, dst)).astype(np.int64)
    x = rng.standard_normal((n, d)).astype(np.float32)
    out = dict(sg_x=x, sg_edge_index=edge_index)
    torch.manual_seed(2)
    conv = ns["GATConv"](d, 10, heads=1, concat=False, dropout=0, add_self_loops=False, bias=False)
    xt = torch.from_numpy(x).requires_grad_(True)
    y, (ei, alpha) = conv(xt, torch.from_numpy(edge_index), return_attention_weights=True)
    dy = torch.from_numpy(rng.standard_normal(y.shape).astype(np.float32))
    y.backward(dy)
    out.update(sg_conv_lin=conv.lin_src.detach().numpy().copy(), sg_conv_att_src=conv.att_src.detach().numpy().copy(),
               sg_conv_att_dst=conv.att_dst.detach().numpy().copy(), sg_conv_out=y.detach().numpy(), sg_conv_alpha=alpha.detach().numpy(),
               sg_conv_dy=dy.numpy(), sg_conv_dx=xt.grad.numpy().copy(), sg_conv_dlin=conv.lin_src.grad.numpy().copy(),
               sg_conv_datt_src=conv.att_src.grad.numpy().copy(), sg_conv_datt_dst=conv.att_dst.grad.numpy().copy())
    torch.manual_seed(3)
    m = Stagate([d, 12, 6], device="cpu")
    for k, v in m.state_dict().items():
        out["sg_sd0::" + k] = v.numpy().copy()
    with torch.no_grad():
        h2, h4 = m(torch.from_numpy(x), torch.from_numpy(edge_index))
    out.update(sg_h2=h2.numpy().copy(), sg_h4=h4.numpy().copy())
    m.pretrain(x, edge_index, lr=1e-2, weight_decay=1e-4, epochs=5, gradient_clipping=5)
    out["sg_rep"] = m.rep
    for k, v in m.state_dict().items():
        out["sg_sd1::" + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "stagate.npz"), **out)
    print("stagate.npz:", len(out), "arrays")


def make_free_riders():
    """free_riders.npz — the layers SURVEY.md §8(f)4 names as re-users of the GCN / attention kernels, each produced by the
    reference's OWN class lifted from /root/reference and run on torch-CPU: scGNN2's ``GraphConvolution`` (scgnn2.py:479-505),
    ``GATLayer`` / ``GAT`` (:883-1189) and ``Graph_AE`` (:373-415); DSTG's ``GraphConvolution`` / ``GCN`` (dstg.py:37-143);
    STdGCN's ``conGraphConvolutionlayer`` (stdgcn.py:63-92); GraphSCI's ``GNNModel`` (graphsci.py:107-124) over the DGL graph stub
    and the restated dgl.nn.GraphConv.  Dropout rates are 0 so every run is deterministic."""
    import types

    import torch.nn as nn
    import torch.nn.functional as F
    rng = np.random.default_rng(2025)
    out = {}
    n, fin, fout = 72, 20, 12
    x = rng.standard_normal((n, fin)).astype(np.float32)
    adj = sp.random(n, n, density=0.1, random_state=3, format="csr", dtype=np.float32)
    adj.data = rng.uniform(0.1, 1.0, adj.nnz).astype(np.float32)
    adj = (adj + sp.eye(n, dtype=np.float32)).tocsr()
    adj.sort_indices()
    dy = rng.standard_normal((n, fout)).astype(np.float32)
    out.update(fr_x=x, fr_dy=dy, fr_adj_indptr=adj.indptr.astype(np.int32), fr_adj_indices=adj.indices.astype(np.int32), fr_adj_data=adj.data)
    a_coo = scipy_to_torch_coo(adj)

    def run_layer(tag, layer, act_on=None):
        for k, v in layer.state_dict().items():
            out[f"{tag}_sd::{k}"] = v.numpy().copy()
        xt = torch.from_numpy(x.copy()).requires_grad_(True)
        y = layer(xt, a_coo)
        y.backward(torch.from_numpy(dy))
        out[f"{tag}_out"] = y.detach().numpy()
        out[f"{tag}_dX"] = xt.grad.numpy().copy()
        for k, p_ in layer.named_parameters():
            out[f"{tag}_grad::{k}"] = p_.grad.numpy().copy()

    s2 = "dance/modules/single_modality/imputation/scgnn2.py"
    GC2 = ref_extract.extract(s2, "GraphConvolution", {"Module": nn.Module})
    torch.manual_seed(11)
    run_layer("s2gc_relu", GC2(fin, fout, 0., act=F.relu))
    run_layer("s2gc_lin", GC2(fin, fout, 0., act=lambda t: t))
    GC_d = ref_extract.extract("dance/modules/spatial/cell_type_deconvo/dstg.py", "GraphConvolution")
    run_layer("dstg_nobias", GC_d(fin, fout, None, bias=False))
    run_layer("dstg_bias", GC_d(fin, fout, None, bias=True))
    dl = ref_extract.extract("dance/modules/spatial/cell_type_deconvo/dstg.py", "dropout_layer", {"sparse_dropout": None})
    GCN_d = ref_extract.extract("dance/modules/spatial/cell_type_deconvo/dstg.py", "GCN", {"GraphConvolution": GC_d, "dropout_layer": dl})
    run_layer("dstg_gcn", GCN_d(fin, 16, fout, bias=False, dropout=0.))
    CG = ref_extract.extract("dance/modules/spatial/cell_type_deconvo/stdgcn.py", "conGraphConvolutionlayer", {"Module": nn.Module})
    run_layer("stdgcn", CG(fin, fout, bias=True))

    # ---- scGNN2 graph attention: one layer (2 heads, concat, skip projection) with gradients, then the 2-layer GAT and Graph_AE
    src, dst = adj.nonzero()[1], adj.nonzero()[0]      # edge j -> i for every stored A[i, j]
    edge_index = np.stack((src, dst)).astype(np.int64)
    out["fr_edge_index"] = edge_index
    GATLayer = ref_extract.extract(s2, "GATLayer")
    GAT = ref_extract.extract(s2, "GAT", {"GATLayer": GATLayer})
    torch.manual_seed(12)
    for tag, kw, width in (("gat_concat", dict(num_of_heads=2, concat=True, activation=nn.ELU()), 2 * 7),
                           ("gat_mean", dict(num_of_heads=3, concat=False, activation=None), 7),
                           ("gat_same", dict(num_of_heads=2, concat=True, activation=None, num_out=fin), 2 * fin)):
        fo = kw.pop("num_out", 7)
        layer = GATLayer(fin, fo, dropout_prob=0.0, add_skip_connection=True, bias=True, log_attention_weights=True, **kw)
        with torch.no_grad():
            layer.bias.copy_(torch.from_numpy(rng.standard_normal(layer.bias.shape).astype(np.float32) * 0.1))
            layer.scoring_fn_source.mul_(4.0)   # spread the scores: the global-max shift then matters
            layer.scoring_fn_target.mul_(4.0)
        for k, v in layer.state_dict().items():
            out[f"{tag}_sd::{k}"] = v.numpy().copy()
        xt = torch.from_numpy(x.copy()).requires_grad_(True)
        y, _ = layer((xt, torch.from_numpy(edge_index)))
        g = torch.from_numpy(rng.standard_normal((n, width)).astype(np.float32))
        y.backward(g)
        out.update({f"{tag}_out": y.detach().numpy(), f"{tag}_dy": g.numpy(), f"{tag}_dX": xt.grad.numpy().copy(),
                    f"{tag}_att": layer.attention_weights.detach().numpy().copy()})
        for k, p_ in layer.named_parameters():
            if p_.grad is not None:
                out[f"{tag}_grad::{k}"] = p_.grad.numpy().copy()
    IPD = ref_extract.extract(s2, "InnerProductDecoder")
    GAE = ref_extract.extract(s2, "Graph_AE", {"GAT": GAT, "GraphConvolution": GC2, "InnerProductDecoder": IPD})
    torch.manual_seed(13)
    gae = GAE(fin, 6, gat_dropout=0, multi_heads=2, gat_hid_embed=8)
    gae.eval()
    for k, v in gae.state_dict().items():
        out[f"gae_sd::{k}"] = v.numpy().copy()
    with torch.no_grad():
        z, _, rec = gae(torch.from_numpy(x), torch.from_numpy(edge_index), use_GAT=True)
        z2, info, rec2 = gae(torch.from_numpy(x), a_coo, use_GAT=False)
    out.update(gae_gat_z=z.numpy(), gae_gat_recon=rec.numpy(), gae_gcn_z=z2.numpy(), gae_gcn_logvar=info[1].numpy(), gae_gcn_recon=rec2.numpy())

    # ---- GraphSCI GNNModel over the DGL stub graph (pattern of adj, edge j -> i) ----------------------------------------------
    GraphConv = ref_extract.dgl_graphconv_stub()
    GNNModel = ref_extract.extract("dance/modules/single_modality/imputation/graphsci.py", "GNNModel",
                                   {"dglnn": types.SimpleNamespace(GraphConv=GraphConv)})
    torch.manual_seed(14)
    gm = GNNModel(in_feats=fin, out_feats=9, dropout=0., n_hidden1=16, n_hidden2=10)
    for k, v in gm.state_dict().items():
        out[f"gsci_sd::{k}"] = v.numpy().copy()
    g = ref_extract.DGLStubGraph(src, dst)
    g.ndata["feat"] = torch.from_numpy(x)
    h = gm.conv2(g, gm.conv1(g, g.ndata["feat"]))
    mean = gm.dec_mean(g, h)
    gdy = torch.from_numpy(rng.standard_normal(mean.shape).astype(np.float32))
    mean.backward(gdy)
    out.update(gsci_h=h.detach().numpy(), gsci_mean=mean.detach().numpy(), gsci_dy=gdy.numpy())
    for k, p_ in gm.named_parameters():
        if p_.grad is not None:
            out[f"gsci_grad::{k}"] = p_.grad.numpy().copy()
    torch.manual_seed(15)
    with torch.no_grad():
        z_adj, log_std, z_mean = gm(g)
    out.update(gsci_fwd_log_std=log_std.numpy(), gsci_fwd_mean=z_mean.numpy())
    np.savez_compressed(os.path.join(HERE, "free_riders.npz"), **out)
    print("free_riders.npz:", len(out), "arrays")


def make_gc_dec():
    """gc_dec.npz — the two-layer DEC model of spagcn.py:588-697 (``GC_DEC``: not instantiated by ``SpaGCN``, which uses SimpleGCDEC),
    lifted from /root/reference with the reference's own GraphConvolution and run on torch-CPU with dropout 0: forward (z, q), the KL
    loss against its own target distribution with the gradients of every parameter, and ``fit_with_init`` (5 SGD epochs)."""
    import logging

    import pandas as pd
    import torch.optim as optim
    spa = "dance/modules/spatial/spatial_domain/spagcn.py"
    GC = ref_extract.extract(spa, "GraphConvolution")
    GCDEC = ref_extract.extract(spa, "GC_DEC", {"GraphConvolution": GC, "optim": optim, "pd": pd, "logger": logging.getLogger("reference")})
    rng = np.random.default_rng(77)
    n, fin, h1, h2, k = 60, 14, 10, 6, 3
    x = rng.standard_normal((n, fin)).astype(np.float32)
    adj = np.exp(-rng.uniform(0, 3, (n, n))).astype(np.float32) * (rng.random((n, n)) < 0.2)
    adj = ((adj + adj.T) / 2 + np.eye(n, dtype=np.float32)).astype(np.float32)
    torch.manual_seed(21)
    m = GCDEC(fin, h1, h2, n_clusters=k, dropout=0.0, alpha=0.2)
    with torch.no_grad():
        m.mu.copy_(torch.from_numpy(rng.standard_normal((k, h2)).astype(np.float32)))
    out = {"gd_x": x, "gd_adj": adj, "gd_dims": np.array([fin, h1, h2, k])}
    for kk, v in m.state_dict().items():
        out[f"gd_sd::{kk}"] = v.numpy().copy()
    z, q = m(torch.from_numpy(x), torch.from_numpy(adj))
    p = m.target_distribution(q).data
    loss = m.loss_function(p, q)
    loss.backward()
    out.update(gd_z=z.detach().numpy(), gd_q=q.detach().numpy(), gd_p=p.numpy(), gd_loss=np.float64(loss.item()))
    for kk, p_ in m.named_parameters():
        out[f"gd_grad::{kk}"] = p_.grad.numpy().copy()
    init_y = rng.integers(0, k, n)
    init_y[:k] = np.arange(k)
    m.zero_grad()
    m.fit_with_init(x, adj, init_y, lr=0.01, epochs=5, update_interval=2, opt="sgd")
    zf, qf = m.predict(x, adj)
    out.update(gd_init_y=init_y, gd_fit_z=zf.detach().numpy(), gd_fit_q=qf.detach().numpy())
    for kk, v in m.state_dict().items():
        out[f"gd_fit_sd::{kk}"] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "gc_dec.npz"), **out)
    print("gc_dec.npz:", len(out), "arrays")


def make_wgc_alpha():
    """wgc_alpha.npz — the reference's own ``WeightedGraphConvAlpha`` (graphsc.py:487-566: per-edge-type learnable alpha, not instantiated
    by the reference's models) lifted from /root/reference and run over the DGL graph stub on torch-CPU: forward and the gradients of
    features, the external weight and alpha, norm "both", tanh activation, a bias."""
    import torch.nn as nn
    gsc = "dance/modules/single_modality/clustering/graphsc.py"

    class GraphConv(nn.Module):  # dgl.nn.pytorch.GraphConv.__init__ / reset_parameters
        def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None, allow_zero_in_degree=False):
            super().__init__()
            self._in_feats, self._out_feats, self._norm, self._allow_zero_in_degree = in_feats, out_feats, norm, allow_zero_in_degree
            self.weight = nn.Parameter(torch.Tensor(in_feats, out_feats)) if weight else None
            self.bias = nn.Parameter(torch.Tensor(out_feats)) if bias else None
            if self.weight is not None:
                nn.init.xavier_uniform_(self.weight)
            if self.bias is not None:
                nn.init.zeros_(self.bias)
            self._activation = activation

    dgl = ref_extract.dgl_stub()
    ns = {"GraphConv": GraphConv, "fn": dgl.function, "DGLError": RuntimeError, "expand_as_pair": lambda feat, g: (feat, feat[:g.number_of_dst_nodes()])}
    WGCA = ref_extract.extract(gsc, "WeightedGraphConvAlpha", ns)
    rng = np.random.default_rng(0)
    n_genes, n_cells, d, h = 5, 7, 6, 4
    n = n_genes + n_cells
    ids = np.concatenate([np.arange(n_genes), -np.ones(n_cells)]).astype(np.int64)   # genes: their id, cells: -1
    dense = (rng.random((n, n)) < 0.35)
    np.fill_diagonal(dense, True)                                                     # dense[v, u]: edge u -> v
    dst, src = np.nonzero(dense)                                                      # CSR order of the destination rows
    g = ref_extract.DGLStubGraph(src, dst)
    g.ndata["id"] = torch.from_numpy(ids)[:, None]  # [N, 1]: with 1-d ids the reference's np.where broadcasts to [E, E]
    feat = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy(rng.standard_normal((d, h)).astype(np.float32)).requires_grad_(True)
    alpha = torch.from_numpy(rng.uniform(0.5, 1.5, (n_genes + 2, 1)).astype(np.float32)).requires_grad_(True)  # [G + 2, 1], as scDeepSort's alpha
    layer = WGCA(d, h, norm="both", weight=False, bias=True, activation=torch.tanh)
    bias = rng.standard_normal(h).astype(np.float32)
    with torch.no_grad():
        layer.bias.copy_(torch.from_numpy(bias))
    out_t = layer(g, feat, weight=w, alpha=alpha, gene_num=n_genes)
    dy = torch.from_numpy(rng.standard_normal((n, h)).astype(np.float32))
    out_t.backward(dy)
    out = dict(wa_ids=ids, wa_dense=dense, wa_feat=feat.detach().numpy(), wa_w=w.detach().numpy(), wa_alpha=alpha.detach().numpy(), wa_bias=bias,
               wa_dy=dy.numpy(), wa_out=out_t.detach().numpy(), wa_dfeat=feat.grad.numpy(), wa_dw=w.grad.numpy(), wa_dalpha=alpha.grad.numpy(),
               wa_dims=np.array([n_genes, n_cells, d, h]))
    np.savez_compressed(os.path.join(HERE, "wgc_alpha.npz"), **out)
    print("wgc_alpha.npz:", len(out), "arrays")


def make_small_transforms():
    """small_transforms.npz — ``FilterGenesMatch.__call__`` (filter.py:417-435), ``FilterCellsType.__call__`` (:1484-1512) and
    ``UpdateSizeFactors.__call__`` (normalize.py:653-659) run as the reference wrote them, on stand-in Data objects (pandas frames, a
    recording ``_inplace_subset_var`` / ``filter_by_mask``): which genes / cells they keep and the size factors they write."""
    import logging
    import types

    import pandas as pd
    import scipy.sparse as sp_
    rng = np.random.default_rng(5)
    out = {}
    names = np.array(["ERCC-1", "Actb", "MT-Co1", "mt-Nd1", "Gapdh-ps", "Xist", "ERCC10", "Malat1-PS"])
    out["st_names"] = names
    call = ref_extract.extract_method("dance/transforms/filter.py", "FilterGenesMatch", "__call__")
    for tag, prefixes, suffixes, cs in (("a", ["ERCC", "MT-"], ["-ps"], False), ("b", ["mt-"], [], True), ("c", [], ["-PS"], True), ("d", [], [], False)):
        kept = {}
        var_names = pd.Index(names)
        data = types.SimpleNamespace(shape=(4, len(names)),
                                     data=types.SimpleNamespace(var_names=var_names, _inplace_subset_var=lambda keep, kept=kept: kept.update(v=np.asarray(keep))))
        if cs:
            prefixes, suffixes = [i.upper() for i in prefixes], [i.upper() for i in suffixes]  # what FilterGenesMatch.__init__ does (:413-415)
        call(types.SimpleNamespace(prefixes=prefixes, suffixes=suffixes, case_sensitive=cs, logger=logging.getLogger("reference")), data)
        out[f"st_match_{tag}_kept"] = np.array(kept["v"], dtype=str)
    call = ref_extract.extract_method("dance/transforms/filter.py", "FilterCellsType", "__call__", {"Data": object})
    lab = np.array([0] * 12 + [1] * 11 + [2] * 10 + [3] * 3)
    rng.shuffle(lab)
    one_hot = pd.DataFrame(np.eye(4)[lab], columns=["a", "b", "c", "d"], index=[str(i) for i in range(len(lab))])
    out["st_type_labels"] = lab
    for thr in (2, 3, 10, 11):
        got = {}
        data = types.SimpleNamespace(data=types.SimpleNamespace(obsm={"cell_type": one_hot}, n_obs=len(lab), obs_names=one_hot.index),
                                     filter_by_mask=lambda m, got=got: got.update(m=np.asarray(m, dtype=bool)))
        call(types.SimpleNamespace(cell_type_threshold=thr), data)
        out[f"st_type_keep_{thr}"] = got["m"]
    call = ref_extract.extract_method("dance/transforms/normalize.py", "UpdateSizeFactors", "__call__", {"Data": object, "sp": sp_})
    x = rng.poisson(2.0, (9, 6)).astype(np.float32)
    out["st_sf_x"] = x
    for tag, mat in (("dense", x), ("sparse", sp_.csr_matrix(x))):
        obs = pd.DataFrame(index=[str(i) for i in range(9)])
        call(types.SimpleNamespace(), types.SimpleNamespace(data=types.SimpleNamespace(X=mat, obs=obs)))
        out[f"st_sf_{tag}_n_counts"], out[f"st_sf_{tag}_size_factors"] = np.asarray(obs["n_counts"], dtype=np.float64), np.asarray(obs["size_factors"], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "small_transforms.npz"), **out)
    print("small_transforms.npz:", len(out), "arrays")


def make_gene_filters():
    """gene_filters.npz — the reference's own ``FilterGenes.__call__`` (filter.py:466-518) with ``FilterGenesPercentile`` /
    ``FilterGenesTopK._get_preserve_mask`` (:583-586, :650-662) and ``FilterGenesCommon.__call__`` (:362-383) run as written on
    stand-in Data objects: the (name-sorted) gene lists they hand to ``_inplace_subset_var``, the summaries and the columns they write."""
    import logging
    import types
    from typing import Dict, Union

    import pandas as pd
    import scipy.sparse as sp_
    rng = np.random.default_rng(22)
    n, g = 60, 40
    lam = rng.gamma(2.0, 3.0, g)    # expressed well enough that no two genes share a statistic (0/1 genes tie in var / mean exactly, and
    x = rng.poisson(lam[None, :] * rng.uniform(0.4, 2.0, (n, 1))).astype(np.float32)   # the reference then decides by fp32 rounding noise)
    x[:, 3] = 0
    x[:30, 5] = 0          # gene 5: silent in the first half of the cells
    x[30:, 9] = 0          # gene 9: silent in the second half
    names = np.array([f"g{i}" for i in range(g)])   # sorted by name != sorted by number
    out = {"gf_x": x, "gf_names": names}
    call = ref_extract.extract_method("dance/transforms/filter.py", "FilterGenes", "__call__", {"sp": sp_, "DevError": RuntimeError})
    pct = ref_extract.extract_method("dance/transforms/filter.py", "FilterGenesPercentile", "_get_preserve_mask")
    topk = ref_extract.extract_method("dance/transforms/filter.py", "FilterGenesTopK", "_get_preserve_mask")
    log = logging.getLogger("reference")

    def run(tag, rule, mode, whitelist=None, **rule_kw):
        var = pd.DataFrame(index=pd.Index(names))
        if whitelist is not None:
            var["keep_me"] = np.isin(np.arange(g), whitelist)
        got = {}
        inner = types.SimpleNamespace(var=var, var_names=var.index, uns={}, _inplace_subset_var=lambda sel: got.update(sel=list(sel)))
        data = types.SimpleNamespace(data=inner, shape=x.shape, get_feature=lambda **kw: x.copy())
        me = types.SimpleNamespace(mode=mode, channel=None, channel_type=None, whitelist_indicators="keep_me" if whitelist is not None else None,
                                   add_n_counts=True, add_n_cells=True, inplace=True, logger=log, **rule_kw)
        me._get_preserve_mask = types.MethodType(rule, me)
        call(me, data)
        out[f"gf_{tag}_selected"] = np.array(got["sel"], dtype=str)
        out[f"gf_{tag}_summary"] = np.asarray(inner.uns["gene_summary"], dtype=np.float64)
        srt = np.sort(out[f"gf_{tag}_summary"])
        assert (np.diff(srt) > 1e-4 * np.abs(srt[1:])).all(), (tag, "tied summaries: pick another seed")   # (gene 3 alone is silent)
        out[f"gf_{tag}_n_counts"], out[f"gf_{tag}_n_cells"] = np.asarray(var["n_counts"], dtype=np.float64), np.asarray(var["n_cells"], dtype=np.int64)

    for mode in ("sum", "var", "cv", "rv"):
        run(f"pct_{mode}", pct, mode, min_val=10, max_val=90)
        run(f"top_{mode}", topk, mode, num_genes=12, top=True)
    run("pct_default", pct, "sum", min_val=1, max_val=99)
    run("bottom_cv", topk, "cv", num_genes=7, top=False)
    run("top_all", topk, "sum", num_genes=99, top=True)
    run("pct_whitelist", pct, "sum", whitelist=[3, 17], min_val=10, max_val=90)

    common = ref_extract.extract_method("dance/transforms/filter.py", "FilterGenesCommon", "__call__", {"DevError": RuntimeError})
    by_split = ref_extract.extract_method("dance/transforms/filter.py", "FilterGenesCommon", "_select_by_splits",
                                          {"Dict": Dict, "Union": Union, "ad": types.SimpleNamespace(AnnData=object)})
    by_batch = ref_extract.extract_method("dance/transforms/filter.py", "FilterGenesCommon", "_select_by_batch",
                                          {"Dict": Dict, "Union": Union, "ad": types.SimpleNamespace(AnnData=object)})

    class Rows:  # the slice of AnnData the method touches: row selection by positions or by obs names, .X, .obs, .var_names
        def __init__(self, mat, obs):
            self.X, self.obs, self.var_names = mat, obs, pd.Index(names)
            self.got = None

        def __getitem__(self, idx):
            pos = self.obs.index.get_indexer(idx) if len(idx) and isinstance(list(idx)[0], str) else np.asarray(list(idx))
            return Rows(self.X[pos], self.obs.iloc[pos])

        def _inplace_subset_var(self, sel):
            self.got = list(sel)

    obs = pd.DataFrame({"batch": np.r_[np.zeros(30, int), np.ones(20, int), np.full(10, 2)]}, index=[f"c{i}" for i in range(n)])
    for tag, mat in (("dense", x), ("sparse", sp_.csr_matrix(x))):
        for mode in ("batch", "split"):
            inner = Rows(mat, obs)
            splits = {"train": list(range(0, 30)), "test": list(range(30, 60)), "val": list(range(25, 35))}
            data = types.SimpleNamespace(data=inner, get_split_idx=lambda k, error_on_miss=True: splits[k])
            me = types.SimpleNamespace(batch_key="batch" if mode == "batch" else None, split_keys=["train", "test"] if mode == "split" else None, logger=log)
            me._select_by_splits, me._select_by_batch = types.MethodType(by_split, me), types.MethodType(by_batch, me)
            common(me, data)
            out[f"gf_common_{mode}_{tag}"] = np.array(inner.got, dtype=str)
    np.savez_compressed(os.path.join(HERE, "gene_filters.npz"), **out)
    print("gene_filters.npz:", len(out), "arrays")


def make_feature_feature_graph():
    """feature_feature_graph.npz — ``FeatureFeatureGraph.__call__`` (feature_feature_graph.py:44-87) run as written, with scipy's
    ``spearmanr`` / ``coo_matrix``, the reference's ``dist_to_rbf`` and a recording stand-in for ``dgl.graph`` /
    ``dgl.nn.EdgeWeightNorm`` (norm="both" on positive weights: w / sqrt(weighted out-degree(src) * weighted in-degree(dst))
    [3P-memory: dgl 1.1.3]): the edge list in the reference's order and the edge weights."""
    import types

    from scipy.sparse import coo_matrix
    from scipy.stats import spearmanr
    rng = np.random.default_rng(31)
    n, g = 90, 36
    base = rng.normal(size=(n, 6))
    x = (base @ rng.normal(size=(6, g)) + 0.8 * rng.normal(size=(n, g))).astype(np.float32)
    x[:, 4] = -x[:, 2] + 0.1 * rng.normal(size=n).astype(np.float32)     # a strongly negative pair
    x[:, 7] = np.round(x[:, 7])                                           # ties for the rank correlation
    out = {"ffg_x": x}

    class Graph:
        def __init__(self, edges, num_nodes):
            self.src, self.dst, self.n = edges[0].long(), edges[1].long(), num_nodes
            self.ndata, self.edata = {}, {}

        def num_edges(self):
            return len(self.src)

    def edge_weight_norm():
        def norm(graph, w):
            out_deg = torch.zeros(graph.n).index_add_(0, graph.src, w)
            in_deg = torch.zeros(graph.n).index_add_(0, graph.dst, w)
            return out_deg[graph.src].pow(-0.5) * in_deg[graph.dst].pow(-0.5) * w
        return norm

    dist_to_rbf = ref_extract.extract("dance/utils/matrix.py", "dist_to_rbf")
    call = ref_extract.extract_method("dance/transforms/graph/feature_feature_graph.py", "FeatureFeatureGraph", "__call__", {
        "spearmanr": spearmanr, "coo_matrix": coo_matrix, "dist_to_rbf": dist_to_rbf, "dgl": types.SimpleNamespace(graph=Graph),
        "dglnn": types.SimpleNamespace(EdgeWeightNorm=edge_weight_norm)})
    cases = {"pearson": dict(threshold=0.3, positive_only=False, score_func="pearson", kw={}),
             "pearson_pos": dict(threshold=0.45, positive_only=True, score_func="pearson", kw={}),
             "pearson_raw": dict(threshold=0.3, positive_only=False, score_func="pearson", kw={}, normalize_edges=False),
             "spearman": dict(threshold=0.3, positive_only=False, score_func="spearman", kw={}),
             "rbf": dict(threshold=0.3, positive_only=False, score_func="rbf", kw={}),
             "rbf_ind": dict(threshold=0.35, positive_only=False, score_func="rbf", kw=dict(scale_mode="ind_med_dist", denom_scale=0.8)),
             "rbf_scale": dict(threshold=0.2, positive_only=False, score_func="rbf", kw=dict(scale_mode="scale", denom_scale=9.0))}
    for tag, c in cases.items():
        inner = types.SimpleNamespace(uns={})
        data = types.SimpleNamespace(data=inner, get_feature=lambda return_type="numpy": x.copy())
        me = types.SimpleNamespace(threshold=c["threshold"], positive_only=c["positive_only"], normalize_edges=c.get("normalize_edges", True),
                                   score_func=c["score_func"], score_func_kwargs=c["kw"], out="FeatureFeatureGraph")
        call(me, data)
        gr = inner.uns["FeatureFeatureGraph"]
        out[f"ffg_{tag}_src"], out[f"ffg_{tag}_dst"] = gr.src.numpy().astype(np.int32), gr.dst.numpy().astype(np.int32)
        out[f"ffg_{tag}_weight"] = gr.edata["weight"].numpy()
        assert np.array_equal(gr.ndata["feat"].numpy(), x.T)
        # the similarity nearest to the threshold, for the record: fp32-vs-fp64 rounding must not be able to move an edge
        if c["score_func"] == "pearson":
            s_ = np.abs(np.corrcoef(x.T))
        elif c["score_func"] == "spearman":
            s_ = np.abs(spearmanr(x, axis=0)[0])
        else:
            s_ = None
        if s_ is not None:
            assert np.abs(s_ - c["threshold"]).min() > 1e-4, (tag, np.abs(s_ - c["threshold"]).min())
    np.savez_compressed(os.path.join(HERE, "feature_feature_graph.npz"), **out)
    print("feature_feature_graph.npz:", len(out), "arrays;", {k: len(out[f"ffg_{k}_src"]) for k in cases})


def make_graphsci():
    """graphsci.npz — the reference's whole ``GraphSCI`` class (graphsci.py:126-560) with its own ``AEModel`` / ``GNNModel``, lifted
    from /root/reference (dgl.nn.GraphConv and the DGL graph as the stubs of oracle/ref_extract.py, no file writes) and run on torch-CPU
    with dropout 0: ``get_loss`` on given tensors, three epochs of ``fit`` (masked), ``predict`` and the three scores; plus
    ``CellwiseMaskData.__call__`` (mask.py:152-290) as written, for both distributions and both mask layouts."""
    import logging
    import types
    from pathlib import Path

    from scipy.sparse import csr_matrix, spmatrix
    from scipy.stats import expon
    path = "dance/modules/single_modality/imputation/graphsci.py"
    rng = np.random.default_rng(41)
    out = {}
    # ---- CellwiseMaskData -----------------------------------------------------------------------------------------------------------
    n, g = 40, 24
    counts = (rng.random((n, g)) < 0.45) * rng.integers(1, 30, (n, g))
    counts[3] = 0
    counts[4, 5:] = 0            # a cell with at most min_gene_counts expressed genes
    counts = counts.astype(np.float32)
    out["gs_counts"] = counts
    call = ref_extract.extract_method("dance/transforms/mask.py", "CellwiseMaskData", "__call__", {"spmatrix": spmatrix})
    probs = ref_extract.extract_method("dance/transforms/mask.py", "CellwiseMaskData", "_get_probs", {"expon": expon})
    for tag, distr, rate, test in (("exp_t", "exp", 0.3, True), ("uni_t", "uniform", 0.25, True), ("exp_v", "exp", 0.1, False), ("all", "uniform", 1.0, True)):
        inner = types.SimpleNamespace(layers={})
        data = types.SimpleNamespace(data=inner, get_feature=lambda return_type="sparse": csr_matrix(counts))
        me = types.SimpleNamespace(distr=distr, mask_rate=rate, seed=7, min_gene_counts=5, add_test_mask=test, logger=logging.getLogger("reference"))
        me._get_probs = types.MethodType(probs, me)
        call(me, data)
        for k in ("train_mask", "valid_mask", "test_mask"):
            out[f"gs_mask_{tag}_{k}"] = inner.layers[k]
    # ---- the model ------------------------------------------------------------------------------------------------------------------
    ns = {"dglnn": types.SimpleNamespace(GraphConv=ref_extract.dgl_graphconv_stub())}
    GNNModel = ref_extract.extract(path, "GNNModel", ns)
    build = ref_extract.extract(path, "buildNetwork")
    Disp, Mean = ref_extract.extract(path, "DispActivation"), ref_extract.extract(path, "MeanActivation")
    Mul = ref_extract.extract(path, "MultiplyLayer")
    AE = ref_extract.extract(path, "AEModel", {"MultiplyLayer": Mul, "buildNetwork": build, "DispActivation": Disp, "MeanActivation": Mean})

    class NoDisk:
        """``Path().resolve() / "graphsci"`` that exists already, so the constructor writes nothing."""
        def __call__(self):
            return self

        def resolve(self):
            return self

        def __truediv__(self, other):
            return self

        def exists(self):
            return True

    GraphSCI = ref_extract.extract(path, "GraphSCI", {"BaseRegressionMethod": object, "Path": NoDisk(), "GNNModel": GNNModel, "AEModel": AE,
                                                      "LogLevel": str})
    GraphSCI.save_model = lambda self: None
    n, g = 30, 12
    lam = rng.gamma(2.0, 1.5, g)
    raw = rng.poisson(lam[None, :] * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    raw[np.arange(n), rng.integers(0, g, n)] += 1
    x = np.log1p(raw)
    corr = np.corrcoef(x.T)
    adj = (np.abs(corr) >= 0.12).astype(np.float32)
    src, dst = np.nonzero(adj)
    deg = adj.sum(1)
    w = (1.0 / np.sqrt(deg[src] * deg[dst])).astype(np.float32)
    mask = rng.random((n, g)) < 0.85
    out.update(gs_x=x, gs_raw=raw, gs_src=src.astype(np.int32), gs_dst=dst.astype(np.int32), gs_w=w, gs_mask=mask)

    class Graph(ref_extract.DGLStubGraph):
        def num_nodes(self):
            return self.number_of_nodes()

        def num_edges(self):
            return len(self._src)

    def graph():
        gr = Graph(src, dst, num_src=g, num_dst=None)
        gr.ndata["feat"] = torch.from_numpy(x.T.copy())
        gr.edata["weight"] = torch.from_numpy(w)
        return gr

    torch.manual_seed(42)
    model = GraphSCI(num_cells=n, num_genes=g, dataset="golden", dropout=0.0, gpu=-1, seed=3)
    for k, v in model.state_dict().items():
        out[f"gs_sd::{k}"] = v.numpy().copy()
    xt, rt = torch.from_numpy(x), torch.from_numpy(raw)
    x_train, raw_train = xt * torch.from_numpy(mask), rt * torch.from_numpy(mask)
    # get_loss on fixed tensors
    model.size_factors = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32))
    t = lambda *shape, pos=False: torch.from_numpy((rng.random(shape) + 0.05 if pos else rng.standard_normal(shape)).astype(np.float32))
    args = dict(z_adj=t(g, g), z_adj_log_std=t(g, g) * 0.1, z_adj_mean=t(g, g), z_exp=t(n, g, pos=True), mean=t(n, g, pos=True), disp=t(n, g, pos=True),
                pi=t(n, g, pos=True) * 0.9)
    out["gs_loss_sf"] = model.size_factors.numpy()
    for k, v in args.items():
        out[f"gs_loss_in_{k}"] = v.numpy()
    losses = model.get_loss(rt, torch.from_numpy(adj), mask=mask, le=1.0, la=0.7, ke=2.0, ka=0.5, **args)
    out["gs_loss_out"] = np.array([float(v) for v in losses])
    # three epochs of fit, predict, scores
    torch.manual_seed(43)
    gr = graph()
    log = []
    orig_train = GraphSCI.train

    def train(self, *a, **k):
        r = orig_train(self, *a, **k)
        log.append([self.train_loss, self.loss_adj, self.loss_exp, self.kl, self.valid_loss])
        return r

    GraphSCI.train = train
    model.fit(x_train, raw_train, gr, mask, le=1, la=1e-2, ke=1e2, ka=1, n_epochs=3, lr=1e-3, weight_decay=1e-6, train_idx=list(range(n - 4)))
    out["gs_fit_log"] = np.array(log)
    out["gs_fit_size_factors"] = model.size_factors.numpy()
    for k, v in model.state_dict().items():
        out[f"gs_fit_sd::{k}"] = v.numpy().copy()
    torch.manual_seed(44)
    imputed = model.predict(x_train, raw_train, gr, mask)
    out["gs_pred"] = imputed.numpy()
    out["gs_scores"] = np.array([model.score(xt, imputed.clone(), ~mask, m, log1p=False) for m in ("RMSE", "PCC", "MRE")] +
                                [model.score(xt, imputed.clone(), ~mask, "RMSE", log1p=True, test_idx=list(range(n - 4, n)))])
    np.savez_compressed(os.path.join(HERE, "graphsci.npz"), **out)
    print("graphsci.npz:", len(out), "arrays; fit log", np.array(log)[:, 0])


def make_scheteronet_split():
    """scheteronet_split.npz — ``set_split`` (scheteronet.py:801-827) of the reference's training script run as written on a stand-in
    AnnData: which class becomes out-of-distribution, the index lists left in ``uns``, the columns written to ``obs``."""
    import types
    from collections import Counter

    import pandas as pd
    path = "dance/modules/single_modality/cell_type_annotation/scheteronet.py"
    get_genename = ref_extract.extract(path, "get_genename")
    set_split = ref_extract.extract(path, "set_split", {"get_genename": get_genename, "Counter": Counter, "pd": pd})
    rng = np.random.default_rng(9)
    n, k, g = 50, 4, 6
    lab = rng.integers(0, k, n)
    lab[:5] = 3
    lab[5:][lab[5:] == 3] = 1          # class 3: 5 cells, the rarest
    x = rng.poisson(1.5, (n, g)).astype(np.float32)

    class Stub:
        def __init__(self):
            self.obs = pd.DataFrame(index=[str(i) for i in range(n)])
            self.var = pd.DataFrame({"symbol": [f"S{i}" for i in range(g)]}, index=[f"g{i}" for i in range(g)])
            self.obsm = {"cell_type": pd.DataFrame(np.eye(k)[lab], index=self.obs.index, columns=[f"t{i}" for i in range(k)])}
            self.uns, self.X = {}, x

        def __getitem__(self, mask):
            assert bool(np.all(mask))
            return self

    ad = Stub()
    perm = rng.permutation(n)
    tr, va, te = perm[:30].tolist(), perm[30:40].tolist(), perm[40:].tolist()
    set_split(types.SimpleNamespace(data=ad), tr, va, te)
    out = dict(ss_labels=lab, ss_x=x, ss_train_in=np.array(tr), ss_val_in=np.array(va), ss_test_in=np.array(te),
               **{f"ss_{key}": np.array(ad.uns[key]) for key in ("train_idx", "val_idx", "test_idx", "ood_idx", "id_idx")},
               ss_cell=np.asarray(ad.obs["cell"]), ss_n_counts=np.asarray(ad.obs["n_counts"], dtype=np.float64),
               ss_gene_name=np.asarray(ad.var["gene_name"], dtype=str), ss_cell_type_is_array=np.array(isinstance(ad.obsm["cell_type"], np.ndarray)))
    # ---- convert_dgl_to_original_format (:155-225) on the DGL stub graph with the index lists set_split just wrote -----------------
    NCDataset = ref_extract.extract(path, "NCDataset")

    class PygData:  # the fields torch_geometric.data.Data keeps for these calls
        def __init__(self, x=None, edge_index=None, y=None):
            self.x, self.edge_index, self.y = x, edge_index, y
            self.num_nodes = x.shape[0]

    conv = ref_extract.extract(path, "convert_dgl_to_original_format", {"NCDataset": NCDataset, "Data": PygData, "dgl": types.SimpleNamespace(DGLGraph=object),
                                                                         "sc": types.SimpleNamespace(AnnData=object)})
    e_src, e_dst = rng.integers(0, n, 120), rng.integers(0, n, 120)
    gs = ref_extract.DGLStubGraph(e_src, e_dst, num_src=n, num_dst=None)
    gs.ndata["feat"], gs.ndata["label"] = torch.from_numpy(x), torch.from_numpy(lab)
    gs.num_nodes = lambda: n
    ind, ood_tr, ood_te, _ = conv(gs, ad, "ref")
    out.update(ss_e_src=e_src, ss_e_dst=e_dst, ss_ind_edge_index=ind.edge_index.numpy(), ss_ind_node_idx=np.array(ind.node_idx),
               ss_ind_split_train=np.array(ind.splits["train"]), ss_ind_split_valid=np.array(ind.splits["valid"]), ss_ind_split_test=np.array(ind.splits["test"]),
               ss_ood_node_idx=np.array(ood_tr.node_idx), ss_ind_num_nodes=np.array(ind.num_nodes), ss_ind_y=ind.y.numpy(),
               ss_same_ood=np.array(ood_tr.node_idx == ood_te.node_idx))
    np.savez_compressed(os.path.join(HERE, "scheteronet_split.npz"), **out)
    print("scheteronet_split.npz:", len(out), "arrays")


if __name__ == "__main__":
    if not ref_extract.available():
        raise SystemExit("reference tree not found: golden vectors can only be generated in the build container")
    torch.manual_seed(0)
    make_gcn_layers()
    make_models()
    make_matrix_known_answers()
    make_graph_builders()
    make_graphsc()
    make_scheteronet()
    make_scdsc_fit()
    make_sctag()
    make_stagate()
    make_free_riders()
    make_gc_dec()
    make_wgc_alpha()
    make_small_transforms()
    make_scheteronet_split()
    make_gene_filters()
    make_feature_feature_graph()
    make_graphsci()
    make_scdeepsort()
