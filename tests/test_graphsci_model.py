"""GraphSCI end to end (dance/modules/single_modality/imputation/graphsci.py:126-560) against what the reference's own class did on the
same inputs (tests/golden/graphsci.npz — ``make_graphsci`` lifts the whole class with its AEModel / GNNModel and runs it on torch-CPU
over the DGL stubs): ``get_loss`` on fixed tensors, three epochs of ``fit`` from the same initial weights and the same torch seed
(the reparameterisation noise comes from torch's CPU generator, so the loop is comparable on CPU tensors only), ``predict`` and the
scores; the entry masks of ``CellwiseMaskData`` bit for bit; and the preprocessing pipeline feeding the model.
CPU: the kernels' torch stand-ins.  ``check_graphsci_loss("cuda")`` / ``check_graphsci_pipeline("cuda")`` are the GPU twins."""
import os

import numpy as np
import pytest
import torch

import cpu_ops

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "graphsci.npz")


@pytest.fixture
def cpu_kernels(monkeypatch):
    from dance_amd import kernels
    for name in cpu_ops.STAND_INS:
        monkeypatch.setattr(kernels, name, getattr(cpu_ops, name))
    return kernels


def _graph(g, device):
    """The gene graph of the golden as FeatureFeatureGraph would leave it: CSR by destination, edge weights, ndata["feat"]."""
    import scipy.sparse as sp
    from dance_amd.graph import CSRGraph
    n = g["gs_x"].shape[1]
    m = sp.csr_matrix((g["gs_w"], (g["gs_dst"], g["gs_src"])), shape=(n, n))
    gr = CSRGraph.from_scipy(m, device=device, symmetric=True)
    gr.ndata = {"feat": torch.from_numpy(g["gs_x"].T.copy()).to(device)}
    gr.edata = {"weight": gr.val}
    return gr


def _model(g, device, prefix="gs_sd::"):
    from dance_amd.modules.single_modality.imputation.graphsci import GraphSCI
    n, k = g["gs_x"].shape
    model = GraphSCI(num_cells=n, num_genes=k, dataset="golden", dropout=0.0, seed=3, device=device)
    sd = {key[len(prefix):]: torch.from_numpy(g[key]) for key in g.files if key.startswith(prefix)}
    assert set(sd) == set(model.state_dict()), "state_dict keys differ from the reference's"
    model.load_state_dict(sd)
    return model


def test_cellwise_mask_data_vs_reference_code():
    from dance_amd.data import AnnDataLite, Data, DeviceArray
    from dance_amd.transforms import CellwiseMaskData
    g = np.load(GOLD)
    counts = g["gs_counts"]
    import scipy.sparse as sp
    for tag, kw in (("exp_t", dict(distr="exp", mask_rate=0.3, add_test_mask=True)), ("uni_t", dict(distr="uniform", mask_rate=0.25, add_test_mask=True)),
                    ("exp_v", dict(distr="exp", mask_rate=0.1)), ("all", dict(distr="uniform", mask_rate=1.0, add_test_mask=True))):
        for slot in (counts.copy(), sp.csr_matrix(counts), DeviceArray(torch.from_numpy(counts.copy()))):
            d = Data(AnnDataLite(slot))
            CellwiseMaskData(seed=7, **kw)(d)
            for k in ("train_mask", "valid_mask", "test_mask"):
                assert np.array_equal(d.data.layers[k], g[f"gs_mask_{tag}_{k}"]), (tag, k)
    m = g["gs_mask_exp_t_train_mask"]
    assert m[3].all() and m[4].all() and not m.all()        # the empty cell and the sparse cell are left alone
    with pytest.raises(ValueError):
        CellwiseMaskData(mask_rate=1.5)
    with pytest.raises(ValueError, match="Unknown distribution"):
        CellwiseMaskData(distr="gauss")._get_probs(np.ones(3))
    assert repr(CellwiseMaskData(seed=1)) == "CellwiseMaskData(distr='exp', mask_rate=0.1, seed=1, min_gene_counts=5, add_test_mask=False)"


def check_graphsci_loss(device, tmp_path, monkeypatch):
    """Deterministic pieces (any device): the five loss terms on fixed tensors, AEModel / get_loss gradients flow, scores."""
    monkeypatch.chdir(tmp_path)
    g = np.load(GOLD)
    model = _model(g, device)
    t = lambda k: torch.from_numpy(g[k]).to(device)
    model.size_factors = t("gs_loss_sf")
    n_genes = g["gs_x"].shape[1]
    adj = torch.zeros(n_genes, n_genes, device=device)
    adj[t("gs_src").long(), t("gs_dst").long()] = 1.0
    args = {k: t(f"gs_loss_in_{k}") for k in ("z_adj", "z_adj_log_std", "z_adj_mean", "z_exp", "mean", "disp", "pi")}
    got = model.get_loss(t("gs_raw"), adj, mask=g["gs_mask"], le=1.0, la=0.7, ke=2.0, ka=0.5, **args)
    assert np.allclose([float(v) for v in got], g["gs_loss_out"], rtol=2e-5), ([float(v) for v in got], g["gs_loss_out"])
    # predict from the trained weights is deterministic apart from the sampled z_adj, which the expression output depends on: compare
    # the deterministic half — the auto-encoder on the golden's generated adjacency is covered by the fit test; here the scores
    model = _model(g, device, prefix="gs_fit_sd::")
    imputed = t("gs_pred")
    xt, mask = t("gs_x"), g["gs_mask"]
    n = len(g["gs_x"])
    got = [model.score(xt, imputed.clone(), ~mask, m, log1p=False) for m in ("RMSE", "PCC", "MRE")]
    got.append(model.score(xt, imputed.clone(), ~mask, "RMSE", log1p=True, test_idx=list(range(n - 4, n))))
    assert np.allclose(got, g["gs_scores"], rtol=1e-5), (got, g["gs_scores"])
    with pytest.raises(ValueError):
        model.score(xt, imputed, ~mask)                      # (sic) the default metric "MSE" is not an allowed one


def test_graphsci_loss_and_scores(cpu_kernels, tmp_path, monkeypatch):
    check_graphsci_loss("cpu", tmp_path, monkeypatch)


def test_graphsci_fit_vs_reference_class(cpu_kernels, tmp_path, monkeypatch, capsys):
    monkeypatch.chdir(tmp_path)
    g = np.load(GOLD)
    model = _model(g, "cpu")
    assert (tmp_path / "graphsci").is_dir()
    xt, rt, mask = torch.from_numpy(g["gs_x"]), torch.from_numpy(g["gs_raw"]), g["gs_mask"]
    x_train, raw_train = xt * torch.from_numpy(mask), rt * torch.from_numpy(mask)
    gr = _graph(g, "cpu")
    n = len(xt)
    log = []
    step = model.train

    def train(*a, **k):
        r = step(*a, **k)
        log.append([model.train_loss, model.loss_adj, model.loss_exp, model.kl, model.valid_loss])
        return r

    model.train = train
    torch.manual_seed(43)
    model.fit(x_train, raw_train, gr, mask, le=1, la=1e-2, ke=1e2, ka=1, n_epochs=3, lr=1e-3, weight_decay=1e-6, train_idx=list(range(n - 4)))
    assert "[Epoch2], train_loss" in capsys.readouterr().out
    log = np.array(log)
    assert np.allclose(log[0], g["gs_fit_log"][0], rtol=1e-5), (log[0], g["gs_fit_log"][0])          # the first forward + loss: the same numbers
    assert np.allclose(log, g["gs_fit_log"], rtol=2e-3), (log, g["gs_fit_log"])  # then Adam steps on gradients that agree to rounding
    assert np.allclose(model.size_factors.numpy(), g["gs_fit_size_factors"], rtol=1e-6)
    # three Adam steps move a parameter by at most ~3 lr either way; where the true gradient is zero (a Linear bias in front of a
    # train-mode BatchNorm) the step's sign is rounding noise, on both sides: such a bias may end up to 6 lr from the reference's
    worst = {}
    for k, v in model.state_dict().items():
        noise_driven = k.startswith("aemodel.") and k.endswith((".1.bias", ".5.bias"))   # the Linear layers of buildNetwork
        follows = k.endswith("running_mean")                                             # ... whose bias the running mean tracks
        ok = np.allclose(v.numpy(), g[f"gs_fit_sd::{k}"], rtol=2e-3, atol=6.5e-3 if noise_driven else (2e-3 if follows else 2e-4))
        if not ok:
            worst[k] = float(np.abs(v.numpy() - g[f"gs_fit_sd::{k}"]).max())
    assert not worst, worst
    assert (tmp_path / "graphsci" / "golden.pt").is_file()
    torch.manual_seed(44)
    imputed = model.predict(x_train, raw_train, gr, mask)
    assert np.allclose(imputed.numpy(), g["gs_pred"], rtol=3e-2, atol=1e-3)
    model.load_model()                                       # the best-validation checkpoint the loop wrote
    assert np.isfinite(model.predict(x_train, raw_train, gr, mask).numpy()).all()


def check_graphsci_pipeline(device, tmp_path, monkeypatch):
    """The reference's step list (graphsci.py:170-199) on a DeviceArray matrix, into a short fit: every slot the model reads exists
    with the right shape, raw follows the gene selection, training lowers the loss it reports."""
    from dance_amd.data import AnnDataLite, Data, DeviceArray
    from dance_amd.graph import CSRGraph
    from dance_amd.modules.single_modality.imputation.graphsci import GraphSCI
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(5)
    n, k = 120, 60
    programme = rng.gamma(2.0, 1.0, (4, k))
    member = rng.integers(0, 4, n)
    raw = rng.poisson(programme[member] * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    raw[:, :4] = 0                                             # never expressed: the min_cells filter removes them
    data = Data(AnnDataLite(DeviceArray(torch.from_numpy(raw).to(device))), train_size=100)
    pipe = GraphSCI.preprocessing_pipeline(min_cells=0.05, threshold=0.2, mask_rate=0.2, seed=2)
    assert [type(t).__name__ for t in pipe.transforms] == ["FilterGenesScanpy", "FilterCellsScanpy", "SaveRaw", "Log1P", "FilterGenesTopK", "UpdateRaw",
                                                          "FeatureFeatureGraph", "CellwiseMaskData", "SetConfig"]
    for t in pipe.transforms:
        if hasattr(t, "device"):
            t.device = device
    pipe(data)
    x, x_raw, graph, mask, valid_mask, test_mask = data.get_x(return_type="default")
    kept = [int(v) for v in data.data.var_names]
    assert x.shape == x_raw.shape == mask.shape == (n, 56) and isinstance(graph, CSRGraph) and graph.n_rows == 56 and sorted(kept) == list(range(4, 60))
    assert np.array_equal(np.asarray(x_raw), raw[:, kept]) and np.allclose(np.asarray(x), np.log1p(raw[:, kept]), rtol=1e-6)
    assert not (valid_mask & test_mask).any() and np.array_equal(~mask, valid_mask | test_mask)
    xt = torch.as_tensor(np.asarray(x)).to(device)
    rt = torch.as_tensor(np.asarray(x_raw)).to(device)
    mt = torch.from_numpy(mask).to(device)
    model = GraphSCI(num_cells=n, num_genes=56, dataset="toy", dropout=0.1, seed=0, device=device)
    torch.manual_seed(0)
    model.fit(xt * mt, rt * mt, graph, mask, le=1, la=1e-9, ke=1e2, ka=1, n_epochs=12, lr=1e-2, weight_decay=1e-6, train_idx=data.train_idx)
    first = model.train_loss
    model.load_model()
    imputed = model.predict(xt * mt, rt * mt, graph, mask)
    assert imputed.shape == (n, 56) and bool(torch.isfinite(imputed).all())
    rmse = model.score(xt, imputed.clone(), ~valid_mask, "RMSE", log1p=False, test_idx=data.test_idx)
    assert np.isfinite(rmse) and np.isfinite(first)


def test_graphsci_pipeline_into_fit(cpu_kernels, tmp_path, monkeypatch):
    check_graphsci_pipeline("cpu", tmp_path, monkeypatch)
