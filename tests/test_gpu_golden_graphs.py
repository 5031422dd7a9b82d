"""GPU: the HIP graph builders / aggregation against tests/golden/graph_builders.npz — outputs of the reference's OWN
methods (lifted by AST, run on torch-CPU; see tests/golden/make_golden.py) — on the same inputs."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "graph_builders.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("norm", [0, 1])
def test_cell_feature_graph_vs_reference_output(cuda_device, gold, norm):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import CellFeatureGraph
    data = Data(AnnDataLite(gold["cfg_x"], obsm={"f": gold["cfg_cell_feat"]}, varm={"f": gold["cfg_gene_feat"]}))
    CellFeatureGraph("f", normalize_edges=bool(norm))(data)
    g = data.data.uns["CellFeatureGraph"]
    tag = f"cfg_norm{norm}_"
    src, dst = g.edges()
    assert np.array_equal(src.cpu().numpy(), gold[tag + "src"]) and np.array_equal(dst.cpu().numpy(), gold[tag + "dst"])
    assert rel_err(g.edata["weight"].cpu().numpy().ravel(), gold[tag + "weight"]) < 1e-6
    assert np.array_equal(g.ndata["cell_id"].cpu().numpy(), gold[tag + "cell_id"])
    assert np.array_equal(g.ndata["feat_id"].cpu().numpy(), gold[tag + "feat_id"])
    assert np.array_equal(g.ndata["features"].cpu().numpy(), gold[tag + "features"])


def _csr_by_dst(src, dst, w, n_dst):
    order = np.argsort(dst, kind="stable")
    rowptr = np.zeros(n_dst + 1, dtype=np.int32)
    rowptr[1:] = np.cumsum(np.bincount(dst, minlength=n_dst))
    return rowptr, src[order].astype(np.int32), w[order].astype(np.float32)


def test_sage_aggregate_vs_reference_output(cuda_device, gold):
    """dh_sage_aggregate_f32 / _bf16 == message_func + fn.mean of the reference (gnn.py:62-90)."""
    from dance_amd import kernels
    n_dst = int(gold["sage_n_dst"])
    rowptr, col, w = _csr_by_dst(gold["sage_src"], gold["sage_dst"], gold["sage_w"], n_dst)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda_device)
    cid = gold["sage_cid"]
    h = np.zeros((cid.size, 8), dtype=np.float32)  # 7 features padded to 8 (the bf16 kernel moves 8 per lane)
    h[:, :7] = gold["sage_h"]
    neigh = kernels.sage_aggregate(t(rowptr), t(col), t(w), t(cid), t(cid[:n_dst]), t(gold["sage_alpha"].ravel()), t(h))
    assert rel_err(neigh.cpu().numpy()[:, :7], gold["sage_neigh"]) < 1e-5
    n16 = kernels.sage_aggregate_bf16(t(rowptr), t(col), t(w), t(cid), t(cid[:n_dst]), t(gold["sage_alpha"].ravel()),
                                      t(h).to(torch.bfloat16), out_dtype=torch.float32)
    assert rel_err(n16.cpu().numpy()[:, :7], gold["sage_neigh"]) < 1e-2


@pytest.mark.parametrize("agg", ["sum", "mean"])
def test_weighted_graph_conv_vs_reference_output(cuda_device, gold, agg):
    """GraphConv(norm='both') with edge weights (graphsc.py:428-484) through the fused SpMM options."""
    from dance_amd import kernels
    n_dst = int(gold["sage_n_dst"])
    src, dst = gold["wgc_src"], gold["wgc_dst"]
    rowptr, col, w = _csr_by_dst(src, dst, gold["wgc_w"], n_dst)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda_device)
    feat = gold["wgc_feat"]
    out_deg = np.maximum(np.bincount(src, minlength=feat.shape[0]), 1).astype(np.float32) ** -0.5
    in_deg = np.maximum(np.bincount(dst, minlength=n_dst), 1).astype(np.float32) ** -0.5
    s = kernels.gemm(t(feat) * t(out_deg)[:, None], t(gold["wgc_weight"]))
    y = kernels.spmm_csr(t(rowptr), t(col), t(w), s, rowscale=t(in_deg), bias=t(gold["wgc_bias"]), act=kernels.ACT_RELU,
                         reduce=kernels.REDUCE_MEAN if agg == "mean" else kernels.REDUCE_SUM)
    assert rel_err(y.cpu().numpy(), gold["wgc_out_" + agg]) < 1e-5


def test_heteronet_and_stagate_vs_reference_output(cuda_device, gold):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import HeteronetGraph, StagateGraph
    assert np.array_equal(HeteronetGraph(knn_num=5).build_graph(gold["het_feats"], knears=5), gold["het_edges"])
    xy = gold["stg_xy"]
    data = Data(AnnDataLite(np.zeros((xy.shape[0], 2), np.float32), obsm={"spatial_pixel": xy}))
    StagateGraph("radius", radius=1.7, out="r")(data)
    StagateGraph("knn", n_neighbors=4, out="k")(data)
    assert np.array_equal(np.asarray(data.data.obsp["r"].todense(), dtype=np.float32), gold["stg_radius"])
    assert np.array_equal(np.asarray(data.data.obsp["k"].todense(), dtype=np.float32), gold["stg_knn"])


def test_spagcn_graph_vs_reference_output(cuda_device, gold):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import SpaGCNGraph
    from oracle import matrix as om
    n = gold["spg_xy"].shape[0]
    data = Data(AnnDataLite(np.zeros((n, 2), np.float32), obsm={"spatial": gold["spg_xy"], "spatial_pixel": gold["spg_xy_pixel"]},
                            uns={"image": gold["spg_img"]}))
    SpaGCNGraph(alpha=float(gold["spg_alpha"]), beta=int(gold["spg_beta"]))(data)
    # the reference hands exactly gold["spg_xyz"] to pairwise_distance (pinned by its own known-answer test)
    assert np.array_equal(data.data.obsp["SpaGCNGraph"], om.pairwise_distance(gold["spg_xyz"], 0))
