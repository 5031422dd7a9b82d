"""Methods the reference's call sites use that round 2 had left out (VERDICT r2 "missing" 3): ``SpaGCN.search_set_res``
(spagcn.py:771-805, called by examples/spatial/spatial_domain/spagcn.py:48), ``SimpleGCDEC.fit_with_init`` (:541-584),
``scHeteroNet.evaluate`` / ``evaluate_detect`` + the OOD measures (scheteronet.py:714-789, 1034-1108).

Each is checked against the REFERENCE'S OWN method lifted out of /root/reference (oracle.ref_extract) and run with our model
object as ``self`` — so the comparison isolates the host logic: same kernels (CPU stand-ins) underneath on both sides."""
import functools
import os
import types

import numpy as np
import pytest
import torch

import cpu_ops
from conftest import rel_err
from oracle import ref_extract

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
needs_ref = pytest.mark.skipif(not ref_extract.available(), reason="/root/reference not present")
SH = "dance/modules/single_modality/cell_type_annotation/scheteronet.py"
SP = "dance/modules/spatial/spatial_domain/spagcn.py"


@pytest.fixture
def cpu_kernels(monkeypatch):
    from dance_amd import kernels
    for name in cpu_ops.STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    return kernels


def _ref_measures():
    from sklearn.metrics import average_precision_score, roc_auc_score
    cs = ref_extract.extract(SH, "stable_cumsum")
    fpr = ref_extract.extract(SH, "fpr_and_fdr_at_recall", {"stable_cumsum": cs})
    return ref_extract.extract(SH, "get_measures", {"fpr_and_fdr_at_recall": fpr, "roc_auc_score": roc_auc_score,
                                                    "average_precision_score": average_precision_score})


@needs_ref
def test_ood_measures_vs_reference():
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import get_measures
    ref = _ref_measures()
    rng = np.random.default_rng(0)
    for n_pos, n_neg, ties in ((50, 70, False), (200, 30, True), (5, 5, True), (300, 300, False)):
        pos, neg = rng.normal(1.0, 1.0, n_pos), rng.normal(0.0, 1.0, n_neg)
        if ties:
            pos, neg = np.round(pos, 1), np.round(neg, 1)
        for level in (0.95, 0.5):
            got, want = get_measures(torch.from_numpy(pos), torch.from_numpy(neg), level), ref(torch.from_numpy(pos), torch.from_numpy(neg), level)
            assert np.allclose(np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64), rtol=0, atol=0), (n_pos, n_neg, ties, level)


def test_ood_measures_known_answers():
    """Separable scores: AUROC = AUPR = 1, no false positives at any recall; identical distributions: fpr@0.95 ~ 0.95."""
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import fpr_and_fdr_at_recall, get_measures
    auroc, aupr, fpr, thr = get_measures(np.arange(10, 20.0), np.arange(0, 10.0))
    assert auroc == 1.0 and abs(aupr - 1.0) < 1e-12 and fpr == 0.0 and 10.0 <= thr <= 11.0
    s = np.linspace(0, 1, 2001)
    lab = (np.arange(2001) % 2).astype(np.int32)
    fpr, _ = fpr_and_fdr_at_recall(lab, s, 0.95)
    assert abs(fpr - 0.95) < 0.01
    with pytest.raises(ValueError):
        fpr_and_fdr_at_recall(np.array([0, 1, 2]), np.array([0.1, 0.2, 0.3]))


def _scheteronet_model():
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import scHeteroNet
    gold = np.load(os.path.join(GOLDEN, "scheteronet.npz"))
    n, d, c, hid = (int(v) for v in gold["sh_dims"])
    ei = torch.from_numpy(gold["sh_edge_index"])
    sd = {k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sh_sd0::")}
    m = scHeteroNet(d, c, ei, n, hid, 2, 0.0, True, "cpu", 1e9)
    m.load_state_dict(sd)
    idx = torch.arange(n)
    ds = types.SimpleNamespace(x=torch.from_numpy(gold["sh_x"]), edge_index=ei, y=torch.from_numpy(gold["sh_y"])[:, None],
                               splits={"train": idx[0::3], "valid": idx[1::3], "test": idx[2::3]}, node_idx=idx[: n // 2])
    ood = types.SimpleNamespace(x=ds.x, edge_index=ei, y=ds.y, node_idx=idx[n // 2:])
    return m, ds, ood


@needs_ref
def test_scheteronet_evaluate_detect_vs_reference_method(cpu_kernels):
    from dance_amd.modules.base import resolve_score_func
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import eval_acc
    m, ds, ood = _scheteronet_model()
    import typing
    ns = {"get_measures": _ref_measures(), "resolve_score_func": resolve_score_func, "Optional": typing.Optional, "Union": typing.Union,
          "Mapping": typing.Mapping, "Any": typing.Any}
    ref_detect = ref_extract.extract_method(SH, "scHeteroNet", "evaluate_detect", ns)
    crit = torch.nn.NLLLoss()
    for ood_arg in (ood, [ood, ood]):
        for use_prop, use_2hop in ((False, False), (True, False), (True, True)):
            kw = dict(return_score=True, dataset="x", T=1.0, use_prop=use_prop, use_2hop=use_2hop, oodprop=2, oodalpha=0.5)
            want = ref_detect(m, ds, ood_arg, crit, eval_acc, "cpu", **kw)
            got = m.evaluate_detect(ds, ood_arg, crit, eval_acc, "cpu", **kw)
            assert len(got[0]) == len(want[0]) == (3 * (2 if isinstance(ood_arg, list) else 1) + 2)
            assert np.allclose([float(v) for v in got[0]], [float(v) for v in want[0]], rtol=1e-6, atol=1e-7)
            assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and np.array_equal(got[3], want[3])
    # evaluate(): appends to results[run], tracks the best validation loss, returns the same list
    results = {0: []}
    res = m.evaluate(ds, ood, crit, eval_acc, display_step=100, run=0, results=results, epoch=1, loss=0.0, dataset="x", T=1.0, use_prop=True,
                     use_2hop=False, oodprop=2, oodalpha=0.5)
    assert results[0] == [res] and float(m.min_loss) == float(res[-1])
    assert m.evaluate_detect(ds, ood, crit, eval_acc, "cpu", False, "x", 1.0, True, False, 2, 0.5)[:3] == res[:3]


def _planted_spots(side=14, d=10, seed=1):
    from oracle import matrix as om
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    dom = (xy[:, 0] >= side / 2).astype(int) + 2 * (xy[:, 1] >= side / 2).astype(int)
    embed = (np.eye(4)[dom] @ rng.standard_normal((4, d)) * 2 + rng.standard_normal((xy.shape[0], d))).astype(np.float32)
    return embed, om.pairwise_distance(xy, 0), dom


@needs_ref
def test_simple_gcdec_fit_with_init_vs_reference_method(cpu_kernels):
    from torch import optim
    from dance_amd.modules.spatial.spatial_domain.spagcn import SimpleGCDEC
    embed, adj, dom = _planted_spots()
    adj_exp = np.exp(-adj**2 / (2 * 1.5**2)).astype(np.float32)
    init_y = np.where(dom == 3, 7, dom)  # non-contiguous labels: groupby sorts them (0, 1, 2, 7)
    ref_fit = ref_extract.extract_method(SP, "SimpleGCDEC", "fit_with_init", {"optim": optim})

    def fresh():
        torch.manual_seed(3)
        m = SimpleGCDEC(embed.shape[1], embed.shape[1], device="cpu")
        m.mu = torch.nn.Parameter(torch.zeros(4, embed.shape[1]))
        return m

    for opt, lr in (("sgd", 0.01), ("admin", 0.005)):
        a, b = fresh(), fresh()
        ref_fit(a, embed, adj_exp, init_y, lr=lr, epochs=6, update_interval=2, opt=opt)
        b.fit_with_init(embed, adj_exp, init_y, lr=lr, epochs=6, update_interval=2, opt=opt)
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and rel_err(vb.numpy(), va.numpy()) < 1e-5, (opt, ka)
    # without an earlier fit the centres parameter is created (the reference would fail on the missing attribute)
    torch.manual_seed(3)
    c = SimpleGCDEC(embed.shape[1], embed.shape[1], device="cpu")
    c.fit_with_init(embed, adj_exp, init_y, epochs=2)
    assert c.mu.shape == (4, embed.shape[1]) and c.n_clusters == 4
    _, q = c.predict(embed, adj_exp)
    assert np.allclose(q.sum(1).numpy(), 1, atol=1e-5)


@needs_ref
def test_spagcn_search_set_res_vs_reference_method(cpu_kernels):
    from dance_amd.modules.spatial.spatial_domain import spagcn
    embed, adj, dom = _planted_spots()
    l = spagcn.SpaGCN(device="cpu").search_l(0.5, adj)
    cpu_model = functools.partial(spagcn.SpaGCN, device="cpu")
    ref_search = ref_extract.extract_method(SP, "SpaGCN", "search_set_res", {"SpaGCN": cpu_model})
    for target, kw in ((4, {}), (2, dict(start=0.6, step=0.2)), (50, dict(max_run=1))):
        a, b = spagcn.SpaGCN(l, device="cpu"), spagcn.SpaGCN(l, device="cpu")
        torch.manual_seed(0)
        want = ref_search(a, (embed, adj), l, target, epochs=3, **kw)
        torch.manual_seed(0)
        got = b.search_set_res((embed, adj), l, target, epochs=3, **kw)
        assert got == want and a.res == b.res, (target, got, want)


def test_scheteronet_set_split_vs_reference_code():
    """``set_split`` of scHeteroNet's training script (scheteronet.py:801-827) against the reference's own function run on a stand-in
    AnnData (tests/golden/scheteronet_split.npz): the out-of-distribution class, the index lists, the columns written."""
    import os
    import numpy as np
    import pandas as pd
    import torch
    from dance_amd import data as dd
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import get_genename, set_split
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scheteronet_split.npz"))
    lab, x = g["ss_labels"], g["ss_x"]
    n, k = len(lab), int(lab.max()) + 1
    idx = [str(i) for i in range(n)]
    for slot in (x.copy(), dd.DeviceArray(torch.from_numpy(x.copy()))):
        ad = dd.AnnDataLite(slot, obs=pd.DataFrame(index=idx), var=pd.DataFrame({"symbol": [f"S{i}" for i in range(x.shape[1])]}, index=[f"g{i}" for i in range(x.shape[1])]),
                            obsm={"cell_type": pd.DataFrame(np.eye(k)[lab], index=idx, columns=[f"t{i}" for i in range(k)])})
        data = dd.Data(ad, train_size="all")
        set_split(data, g["ss_train_in"].tolist(), g["ss_val_in"].tolist(), g["ss_test_in"].tolist())
        for key in ("train_idx", "val_idx", "test_idx", "ood_idx", "id_idx"):
            assert list(ad.uns[key]) == g[f"ss_{key}"].tolist(), key
        assert np.array_equal(np.asarray(ad.obs["cell"]), g["ss_cell"]) and np.allclose(np.asarray(ad.obs["n_counts"], dtype=np.float64), g["ss_n_counts"])
        assert [str(v) for v in ad.var["gene_name"]] == [str(v) for v in g["ss_gene_name"]]
        assert isinstance(ad.obsm["cell_type"], np.ndarray) == bool(g["ss_cell_type_is_array"])
    assert list(get_genename(dd.AnnDataLite(x, var=pd.DataFrame({"gene_id": ["a"] * x.shape[1]})))) == ["a"] * x.shape[1]


def test_scheteronet_convert_vs_reference_code():
    """``convert_dgl_to_original_format`` (scheteronet.py:155-225) on the graph object HeteronetGraph produces vs the reference's own
    function over the DGL stub graph: edge_index in edge-id order, the in-distribution dataset's fields and splits, the OOD node lists."""
    import os
    import types
    import numpy as np
    import torch
    from dance_amd.cellgraph import CellGeneGraph
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import NCDataset, OODData, convert_dgl_to_original_format
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scheteronet_split.npz"))
    src, dst = torch.from_numpy(g["ss_e_src"]), torch.from_numpy(g["ss_e_dst"])
    n = len(g["ss_labels"])
    # CSR by destination with edge ids = positions in the edge list (what HeteronetGraph builds, heteronet_graph.py)
    order = torch.argsort(dst * (len(src) + 1) + torch.arange(len(src)))
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n), 0)
    cg = CellGeneGraph(rowptr.to(torch.int32), src[order].to(torch.int32), torch.ones(len(src)), order.to(torch.int32), n,
                       {"feat": torch.from_numpy(g["ss_x"]), "label": torch.from_numpy(g["ss_labels"])})
    adata = types.SimpleNamespace(uns={k: g[f"ss_{k}"].tolist() for k in ("train_idx", "val_idx", "test_idx", "ood_idx", "id_idx")})
    ind, ood_tr, ood_te, back = convert_dgl_to_original_format(cg, adata, "ref")
    assert isinstance(ind, NCDataset) and isinstance(ood_tr, OODData) and back is adata
    assert np.array_equal(ind.edge_index.numpy(), g["ss_ind_edge_index"]) and ind.num_nodes == int(g["ss_ind_num_nodes"])
    assert list(ind.node_idx) == g["ss_ind_node_idx"].tolist() and np.array_equal(ind.y.numpy(), g["ss_ind_y"])
    for ours, key in (("train", "train"), ("valid", "valid"), ("test", "test")):
        assert list(ind.splits[ours]) == g[f"ss_ind_split_{key}"].tolist()
    assert list(ood_tr.node_idx) == g["ss_ood_node_idx"].tolist() and list(ood_te.node_idx) == list(ood_tr.node_idx)
    assert ood_tr.num_nodes == n and torch.equal(ood_tr.edge_index, ind.edge_index) and torch.equal(ind.graph["node_feat"], ind.x)


@needs_ref
@pytest.mark.parametrize("tag", ["mb", "peak"])
def test_reference_scdeepsort_methods_run_on_our_objects(cpu_kernels, tmp_path, monkeypatch, tag):
    """The reference's own ``cal_loss`` / ``evaluate`` / ``predict_proba`` / ``predict`` bodies (scdeepsort.py:208-349) run LIVE with
    our ScDeepSort as ``self``, our graph, and our ``DataLoader`` / blocks standing where dgl's are (``graph_sampler=`` keyword,
    ``enable_cpu_affinity``, ``b.to(device)``, ``num_dst_nodes``): they must return what our own methods return — and what the
    golden file holds for the trained model."""
    from contextlib import nullcontext

    import scdeepsort_golden_checks as chk
    from dance_amd import cellgraph
    SD = "dance/modules/single_modality/cell_type_annotation/scdeepsort.py"
    gold, kw = chk.load()
    m, g, log = chk.fit_case(gold, kw, tag, "cpu", tmp_path, monkeypatch, block_eval=True)
    ns = {"dgl": types.SimpleNamespace(DGLGraph=object), "DataLoader": cellgraph.DataLoader, "NeighborSampler": cellgraph.NeighborSampler,
          "nullcontext": nullcontext, "print": lambda *a, **k: None}
    ref = {name: ref_extract.extract_method(SD, "ScDeepSort", name, ns) for name in ("cal_loss", "evaluate", "predict_proba", "predict")}
    n_genes, n_cells = gold["x"].shape[1], gold["x"].shape[0]
    idx = torch.arange(n_genes, n_genes + n_cells)
    gg = g.to("cpu")
    gg.ndata["label"] = torch.cat((-torch.ones(n_genes, dtype=torch.long), torch.from_numpy(gold["labels"])))
    # evaluate: the per-row .item() loop of the reference vs our vectorised form, on the trained model
    want = ref["evaluate"](m, gg, idx)
    m.full_graph_eval = False
    assert tuple(m.evaluate(gg, idx)) == tuple(want)
    m.full_graph_eval = True
    assert tuple(m.evaluate(gg, idx)) == tuple(want)
    # predict_proba / predict
    prob_ref = ref["predict_proba"](m, gg)
    assert np.abs(prob_ref - m.predict_proba(gg)).max() < 1e-6 and np.abs(prob_ref - gold[f"{tag}_prob"]).max() < 2e-4
    m.predict_proba = lambda graph: prob_ref          # the reference's predict calls self.predict_proba
    for rate, suffix in ((2.0, ""), (1.4, "_loose")):
        pred, unsure = ref["predict"](m, gg, unsure_rate=rate, return_unsure=True)
        assert np.array_equal(pred, gold[f"{tag}_pred{suffix}"]) and np.array_equal(unsure, gold[f"{tag}_unsure{suffix}"])
    del m.predict_proba
    # cal_loss: one more epoch by the reference's body and by ours from the same state, same loader permutation
    import copy
    state = copy.deepcopy(m.model.state_dict())
    opt_state = copy.deepcopy(m.optimizer.state_dict())
    train_idx = idx[: n_cells // 2]
    m.shuffle_generator = torch.Generator().manual_seed(5)
    ours = m.cal_loss(gg, train_idx)
    after_ours = copy.deepcopy(m.model.state_dict())
    m.model.load_state_dict(state)
    m.optimizer.load_state_dict(opt_state)
    gen = torch.Generator().manual_seed(5)
    monkeypatch.setitem(ns, "DataLoader", functools.partial(cellgraph.DataLoader, generator=gen))
    ref_cal = ref_extract.extract_method(SD, "ScDeepSort", "cal_loss", ns)
    theirs = ref_cal(m, gg, train_idx)
    assert abs(ours - theirs) < 1e-5 * abs(theirs)
    for k, v in m.model.state_dict().items():
        assert rel_err(v.numpy(), after_ours[k].numpy()) < 1e-5, k
