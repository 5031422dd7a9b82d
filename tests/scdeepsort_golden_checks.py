"""Shared body of the ScDeepSort golden tests (GPU: tests/test_gpu_scdeepsort_golden.py through the HIP kernels; CPU:
tests/test_models_host_logic.py through the stand-ins of tests/cpu_ops.py).

tests/golden/scdeepsort.npz holds what the REFERENCE'S OWN ``ScDeepSort.fit / cal_loss / evaluate / predict_proba / predict``
(scdeepsort.py:142-349, with its GNN and AdaptiveSAGE) produced on torch-CPU over the DGL stub: per-epoch losses, the
(correct, unsure, accuracy) triples of ``evaluate`` on the training and validation cells, the best-validation checkpoint ``fit``
ends with, the class probabilities and the predictions with their "unsure" flags (tests/golden/make_golden.py::make_scdeepsort).
The product is driven the same way: same initial weights (the reference's, loaded into the freshly built GNN), every random
permutation (split, loader orders) from one generator with the golden's seed."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "scdeepsort.npz")


def load():
    gold = np.load(GOLD)
    return gold, json.loads(str(gold["kw"]))


def build_graph(gold, device):
    """The golden inputs through the product's own CellFeatureGraph transform (normalize_edges=True, the pipeline default)."""
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import CellFeatureGraph
    data = Data(AnnDataLite(gold["x"], obsm={"f": gold["cell_feat"]}, varm={"f": gold["gene_feat"]}))
    t = CellFeatureGraph("f")
    if hasattr(t, "device"):
        t.device = device
    t(data)
    return data.data.uns["CellFeatureGraph"]


def fit_case(gold, kw, tag, device, tmp_path, monkeypatch, *, block_eval=True):
    """Run the product's fit on case ``tag``; returns (model, log) with log["cal_loss"] / log["evaluate"] = every returned value."""
    from dance_amd.modules.single_modality.cell_type_annotation import scdeepsort as mod
    sd0 = {k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f"{tag}_sd0::")}
    real_gnn = mod.GNN

    def gnn_with_reference_init(*a, **k):
        m = real_gnn(*a, **k)
        assert sorted(sd0) == sorted(m.state_dict())  # the reference's parameter names
        m.load_state_dict(sd0)
        return m
    monkeypatch.setattr(mod, "GNN", gnn_with_reference_init)
    log = {"cal_loss": [], "evaluate": []}

    class Recorded(mod.ScDeepSort):
        def cal_loss(self, *a, **k):
            r = super().cal_loss(*a, **k)
            log["cal_loss"].append(r)
            return r

        def evaluate(self, *a, **k):
            r = super().evaluate(*a, **k)
            log["evaluate"].append(r)
            return r
    case = kw["cases"][tag]
    m = Recorded(gold["gene_feat"].shape[1], kw["hid"], 1, "golden", "scds", batch_size=case["batch_size"], device=device, save_root=tmp_path,
                 verbose=False)
    m.shuffle_generator = torch.Generator().manual_seed(kw["seed_order"])
    m.full_graph_eval = not block_eval
    g = build_graph(gold, device)
    m.fit(g, torch.from_numpy(gold["labels"]), epochs=case["epochs"], lr=case["lr"], weight_decay=0, val_ratio=0.2)
    return m, g, log


def check_case(gold, kw, tag, device, tmp_path, monkeypatch, *, block_eval=True, rel_err=None):
    m, g, log = fit_case(gold, kw, tag, device, tmp_path, monkeypatch, block_eval=block_eval)
    # cal_loss: size-weighted mean of the summed cross entropy per batch (scdeepsort.py:247-250)
    assert np.allclose(log["cal_loss"], gold[f"{tag}_losses"], rtol=2e-4, atol=0), (log["cal_loss"], gold[f"{tag}_losses"])
    # evaluate(train), evaluate(val) per epoch: correct / unsure counts (the raw-logit rule of :280-281) and accuracy
    ev = np.array([[c, u, a] for c, u, a in log["evaluate"]], dtype=np.float64)
    assert ev.shape == gold[f"{tag}_eval"].shape
    assert np.array_equal(ev[:, :2], gold[f"{tag}_eval"][:, :2]), (ev, gold[f"{tag}_eval"])
    assert np.allclose(ev[:, 2], gold[f"{tag}_eval"][:, 2], rtol=0, atol=1e-12)
    # fit ends on the best-validation checkpoint (:194-206; "<=" keeps the LAST of equally good epochs)
    sd = m.model.state_dict()
    for k in gold.files:
        if k.startswith(f"{tag}_sd1::"):
            assert rel_err(sd[k.split("::", 1)[1]].detach().cpu().numpy(), gold[k]) < 1e-3, k
    prob = m.predict_proba(g)
    assert prob.shape == gold[f"{tag}_prob"].shape and np.abs(prob - gold[f"{tag}_prob"]).max() < 2e-4
    for rate, suffix in ((2.0, ""), (1.4, "_loose")):
        pred, unsure = m.predict(g, unsure_rate=rate, return_unsure=True)
        assert np.array_equal(unsure, gold[f"{tag}_unsure{suffix}"]), rate
        assert np.array_equal(pred, gold[f"{tag}_pred{suffix}"]), rate
    if tag != "peak":  # the rule is exercised both ways (at lr 0.1 the model is confident about everything)
        for suffix in ("", "_loose"):
            assert gold[f"{tag}_unsure{suffix}"].any() and not gold[f"{tag}_unsure{suffix}"].all()
    return m, g
