"""GPU: the persistent mini-batch steps (csrc/ministep.hip behind dance_amd/ministep.py: dh_graphsc_steps, dh_scdeepsort_steps) against
oracle/ministep.py — a float64 restatement of one batch of the reference's loops (graphsc.py:196-219, scdeepsort.py:238-246) in the
reference's operation order with the kernels' Philox masks made explicit, itself pinned to the reference's own ``fit`` goldens
(tests/test_oracle_ministep.py) — and through ``GraphSC.fit`` / ``ScDeepSort.fit`` against those goldens and the general loop."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import rel_err
from oracle import ministep as oms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "graphsc.npz")


def _graph(n_cells, n_genes, d, seed, dev, density=0.2, normalize_edges=True):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import CellFeatureGraph
    rng = np.random.default_rng(seed)
    x = ((rng.random((n_cells, n_genes)) < density) * rng.integers(1, 9, (n_cells, n_genes))).astype(np.float32)
    data = Data(AnnDataLite(x, obsm={"f": rng.standard_normal((n_cells, d)).astype(np.float32)},
                            varm={"f": rng.standard_normal((n_genes, d)).astype(np.float32)}))
    CellFeatureGraph("f", normalize_edges=normalize_edges)(data)
    return data.data.uns["CellFeatureGraph"]


def _host(g):
    return g.rowptr.cpu().numpy().astype(np.int64), g.col.cpu().numpy().astype(np.int64), g.val.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("n,p,seed,step,sid", [(1000, 0.1, 1, 0, 0), (4097, 0.5, 2**40 + 12345, 3, 4), (3, 0.25, 99, 2**33 + 7, 5), (70001, 0.9, 7, 11, 2),
                                               (513, 0.0, 5, 5, 1)])
def test_dropout_mask_kernel_equals_oracle(cuda_device, n, p, seed, step, sid):
    """The in-kernel Philox4x32-10 draw == the numpy restatement that passes the Random123 known-answer vectors, bit for bit."""
    from dance_amd.ministep import dropout_mask
    got = dropout_mask(n, p, seed, step, sid, cuda_device).cpu().numpy()
    assert np.array_equal(got.astype(np.float64), oms.dropout_mask(n, p, seed, step, sid))


def _gsc_model(f, h, e, agg, p, pd, dev, seed=0):
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    torch.manual_seed(seed)
    m = GraphSC(agg=agg, in_feats=f, hidden_dim=h, hidden_1=e, dropout=p, n_clusters=3, device="cuda")
    m.model.decoder.dropout = pd
    with torch.no_grad():  # biases away from zero, weights large enough for a live ReLU pattern
        m.model.layer1.bias.uniform_(-0.1, 0.1)
        m.model.encoder[0].bias.uniform_(-0.1, 0.1)
    m.model.train()
    return m


def _gsc_params(model):
    return {"W1": model.layer1.weight, "b1": model.layer1.bias, "W2": model.encoder[0].weight, "b2": model.encoder[0].bias}


@pytest.mark.parametrize("b,n_genes,f,h,e,p,pd,agg", [(16, 12, 6, 10, 5, 0.0, 0.0, "sum"), (32, 70, 50, 200, 300, 0.1, 0.1, "sum"),
                                                      (33, 40, 7, 13, 9, 0.3, 0.2, "mean"), (128, 90, 100, 260, 300, 0.1, 0.1, "sum"),
                                                      (2, 5, 1, 1, 1, 0.0, 0.5, "sum")])
def test_graphsc_steps_vs_oracle(cuda_device, b, n_genes, f, h, e, p, pd, agg):
    """Three consecutive steps of dh_graphsc_steps: the collected embedding (first forward), the loss (second forward, fresh dropout, the
    decoder's own dropout), the gradients (phase 1) and the Adam-updated parameters / moments, against the float64 oracle run with the
    same masks — odd widths, widths past one wavefront / one workgroup pass, both aggregations."""
    from dance_amd.ministep import GraphSCStepper
    g = _graph(max(4 * b, 64), n_genes, f, 1, cuda_device, normalize_edges=False)
    rowptr, col, val = _host(g)
    feats = g.ndata["features"].cpu().numpy()
    m = _gsc_model(f, h, e, agg, p, pd, cuda_device)
    optim = torch.optim.Adam(m.model.parameters(), lr=1e-2, fused=True)
    assert GraphSCStepper.eligible(m.model, g, b, optim)
    st = GraphSCStepper(m.model, g, b, optim)
    n_steps = 3
    seeds = (n_genes + torch.randperm(g.number_of_nodes() - n_genes, generator=torch.Generator().manual_seed(3))[:n_steps * b]).to(cuda_device)
    # gradients of the first step alone (the data-parallel phase 1), without touching the parameters
    st.grads = torch.zeros(sum(q.numel() for q in st.params), device=cuda_device)
    st.cfg.grads, st.cfg.phase = st.grads.data_ptr(), 1
    z1, l1 = torch.empty((b, e), device=cuda_device), torch.empty(1, device=cuda_device)
    st.cfg.seeds, st.cfg.z_out, st.cfg.loss_out = seeds.data_ptr(), z1.data_ptr(), l1.data_ptr()
    st.cfg.dropout = p
    st._run(__import__("dance_amd._lib", fromlist=["load"]).load().dh_graphsc_steps, "graphsc_steps", 0, 1)
    params = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in _gsc_params(m.model).items()}
    loss0, emb0, grads0 = oms.graphsc_step(params, rowptr, col, val, feats, n_genes, seeds[:b].cpu().numpy(), dropout=p, decoder_dropout=pd,
                                           seed=st.cfg.seed, step=0, agg=agg)
    flat = st.grads.cpu().numpy()
    off = 0
    for k in ("W1", "b1", "W2", "b2"):
        got = flat[off:off + grads0[k].size].reshape(grads0[k].shape)
        off += grads0[k].size
        assert rel_err(got, grads0[k]) < 1e-4, k
    assert abs(float(l1) - loss0) < 1e-5 * abs(loss0) and rel_err(z1.cpu().numpy(), emb0) < 1e-5
    for t_ in optim.state.values():  # phase 1 ticked the step counters once: back to a fresh optimiser
        t_["step"].zero_()
    # whole steps
    st.cfg.phase, st.cfg.grads, st.grads = 0, None, None
    z, loss = torch.empty((n_steps * b, e), device=cuda_device), torch.empty(n_steps, device=cuda_device)
    st.run(seeds, n_steps, z, loss)
    st.check_flags("test")
    mom = {k: np.zeros_like(v) for k, v in params.items()}
    var = {k: np.zeros_like(v) for k, v in params.items()}
    for s in range(n_steps):
        lo, emb, grads = oms.graphsc_step(params, rowptr, col, val, feats, n_genes, seeds[s * b:(s + 1) * b].cpu().numpy(), dropout=p,
                                          decoder_dropout=pd, seed=st.cfg.seed, step=s, agg=agg)
        assert abs(float(loss[s]) - lo) < 2e-4 * abs(lo), s
        assert rel_err(z[s * b:(s + 1) * b].cpu().numpy(), emb) < 1e-3, s
        for k in params:
            pn, mn, vn = oms.adam_update(torch.from_numpy(params[k]), torch.from_numpy(grads[k]), torch.from_numpy(mom[k]), torch.from_numpy(var[k]),
                                         s + 1, 1e-2, 0.9, 0.999, 1e-8)
            params[k], mom[k], var[k] = pn.numpy(), mn.numpy(), vn.numpy()
    for k, q in _gsc_params(m.model).items():
        assert rel_err(q.detach().cpu().numpy(), params[k]) < 2e-3, k
        assert rel_err(optim.state[q]["exp_avg"].cpu().numpy(), mom[k]) < 1e-3, k
        assert float(optim.state[q]["step"]) == n_steps
    assert st.cfg.step0 == n_steps


def test_graphsc_steps_flag_bad_seeds(cuda_device):
    """A gene id among the seeds (not a cell row) raises through the flag word instead of computing something else."""
    from dance_amd.ministep import GraphSCStepper
    g = _graph(64, 12, 6, 2, cuda_device)
    m = _gsc_model(6, 10, 5, "sum", 0.0, 0.0, cuda_device)
    optim = torch.optim.Adam(m.model.parameters(), lr=1e-3, fused=True)
    st = GraphSCStepper(m.model, g, 8, optim)
    seeds = torch.arange(12, 20, device=cuda_device)
    seeds[3] = 2
    with pytest.raises(RuntimeError, match="not a cell row"):
        st.run(seeds, 1, torch.empty((8, 5), device=cuda_device), torch.empty(1, device=cuda_device))
        st.check_flags("test")


def test_graphsc_aggregate_phase_flags_bad_seeds(cuda_device):
    """The large-batch counters (LDS histograms, seeds fetched 64 at a time per wave) raise the same flags as the per-seed walk: a gene id
    among the seeds; a seed whose row has no self loop."""
    from dance_amd.ministep import GraphSCStepper
    g = _graph(1300, 40, 6, 2, cuda_device)
    m = _gsc_model(6, 10, 5, "sum", 0.0, 0.0, cuda_device)
    optim = torch.optim.Adam(m.model.parameters(), lr=1e-3, fused=True)
    st = GraphSCStepper(m.model, g, 1100, optim)
    seeds = (40 + torch.arange(1100)).to(cuda_device)
    st.aggregate(seeds)
    st.check_flags("test")  # clean
    bad = seeds.clone()
    bad[777] = 3
    with pytest.raises(RuntimeError, match="not a cell row"):
        st.aggregate(bad)
        st.check_flags("test")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _gold_graph(gold):
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.graph import CellFeatureGraph
    data = Data(AnnDataLite(gold["gsc_x"], obsm={"f": gold["gsc_cell_feat"]}, varm={"f": gold["gsc_gene_feat"]}))
    CellFeatureGraph("f", normalize_edges=False)(data)
    return data.data.uns["CellFeatureGraph"]


def _gold_model(gold, tag, agg):
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    kw = json.loads(str(gold["gsc_kw"]))
    kw["agg"] = agg
    m = GraphSC(**kw, n_clusters=3, device="cuda")
    m.model.decoder.dropout = 0.0
    m.model.load_state_dict({k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f"gsc_{tag}_sd0::")})
    return m


@pytest.mark.parametrize("tag,agg,mode", [("mb", "sum", "ministep"), ("mean", "mean", "ministep"), ("mb", "sum", "aggfirst"), ("mean", "mean", "aggfirst")])
def test_graphsc_fit_ministep_vs_reference(cuda_device, gold, monkeypatch, tag, agg, mode):
    """GraphSC.fit on the persistent step (every full batch of an epoch in one C call, the short last batch eagerly) — and on its
    large-batch form ("aggfirst": the aggregation of dh_graphsc_steps phase 3, dense layers / all-pairs decoder / Adam on the big-tile
    kernels; forced at this toy size) — reproduces the reference's OWN fit: every per-batch loss, the final embedding in cell order and
    the final weights (graphsc.npz) — and the general loop's numbers to rounding."""
    from dance_amd.modules.single_modality.clustering import graphsc
    res = {}
    for on in (True, False):
        monkeypatch.setattr(graphsc, "MINISTEP", on)
        monkeypatch.setattr(graphsc, "HIPGRAPH", False)
        monkeypatch.setattr(graphsc, "MINISTEP_MAX_BATCH", 512 if mode == "ministep" else 8)
        g = _gold_graph(gold)
        m = _gold_model(gold, tag, agg)
        m.shuffle_generator = torch.Generator().manual_seed(123)
        m.fit(g, epochs=3, lr=1e-2, batch_size=16)
        assert m.step_mode == (mode if on else "eager")
        res[on] = (np.asarray(m.losses), m.get_latent().copy(), {k: v.detach().cpu().numpy().copy() for k, v in m.model.state_dict().items()})
    ref = gold[f"gsc_{tag}_losses"]
    assert len(res[True][0]) == len(ref) and np.allclose(res[True][0], ref, rtol=2e-4, atol=0)
    assert rel_err(res[True][1], gold[f"gsc_{tag}_z"]) < 1e-3
    for k in gold.files:
        if k.startswith(f"gsc_{tag}_sd1::"):
            assert rel_err(res[True][2][k.split("::", 1)[1]], gold[k]) < 1e-3, k
    assert np.allclose(res[True][0], res[False][0], rtol=1e-5) and rel_err(res[True][1], res[False][1]) < 1e-5


@pytest.mark.parametrize("b,n_cells,n_genes,f,p,agg", [(40, 200, 30, 50, 0.2, "sum"), (1024, 1500, 300, 50, 0.2, "sum"), (1056, 1200, 130, 7, 0.0, "mean"),
                                                       (1024, 1100, 257, 64, 0.1, "sum"), (1030, 1100, 2100, 8, 0.0, "sum")])
def test_graphsc_aggregate_phase_vs_oracle(cuda_device, monkeypatch, b, n_cells, n_genes, f, p, agg):
    """dh_graphsc_steps phase 3: the aggregated layer input of both forwards (own dropout draws) == the oracle's A_norm (X o mask) — small
    and large batches (the LDS-histogram counters from 1024 seeds on), even / odd widths, a ragged last row tile, a gene count that is no
    multiple of the matrix-core form's K chunk, rows of more than 256 entries (the counters fetch a row's column ids 256 at a time).
    tests/test_gpu_ministep.py::test_graphsc_aggregate_mfma_form runs the dense-product form."""
    from dance_amd.ministep import GraphSCStepper
    g = _graph(n_cells, n_genes, f, 4, cuda_device, density=0.15, normalize_edges=False)
    rowptr, col, val = _host(g)
    feats = g.ndata["features"].cpu().numpy().astype(np.float64)
    m = _gsc_model(f, 20, 12, agg, p, 0.0, cuda_device)
    optim = torch.optim.Adam(m.model.parameters(), lr=1e-2, fused=True)
    st = GraphSCStepper(m.model, g, b, optim)
    seeds = (n_genes + torch.randperm(n_cells, generator=torch.Generator().manual_seed(1))[:b]).to(cuda_device)
    ax = st.aggregate(seeds).cpu().numpy()
    st.check_flags("test")
    sd = seeds.cpu().numpy()
    cnt = np.zeros(n_genes)
    for v in sd:
        cs_ = col[rowptr[v]:rowptr[v + 1]]
        np.add.at(cnt, cs_[cs_ < n_genes], 1)
    for k in range(2):
        m_self = oms.dropout_mask(b * f, p, st.cfg.seed, 0, oms.SID_SELF + k).reshape(b, f)
        m_gene = oms.dropout_mask(n_genes * f, p, st.cfg.seed, 0, oms.SID_GENE + k).reshape(n_genes, f)
        xg = feats[:n_genes] * m_gene / np.sqrt(np.maximum(cnt, 1.0))[:, None]
        ref = np.zeros((b, f))
        for i, v in enumerate(sd):
            s_, t_ = rowptr[v], rowptr[v + 1]
            cs_, ws_ = col[s_:t_], val[s_:t_]
            gm = cs_ < n_genes
            ref[i] = ws_[gm] @ xg[cs_[gm]] + ws_[~gm].sum() * feats[v] * m_self[i]
            deg = max(t_ - s_, 1)
            ref[i] /= np.sqrt(deg) * (deg if agg == "mean" else 1.0)
        assert rel_err(ax[k], ref) < 1e-5, k
    assert st.cfg.step0 == 1
    if b >= 1024:
        assert st.cfg.max_row_entries > 0


def test_graphsc_aggregate_mfma_form():
    """The dense-product form of phase 3 (DANCE_AMD_GRAPHSC_AGG=mfma, read once per process: a subprocess) == the default gather form."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np, torch\n"
        "sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))\n"
        "import test_gpu_ministep as t\n"
        "from dance_amd.ministep import GraphSCStepper\n"
        "dev = torch.device('cuda')\n"
        "out = []\n"
        "for (b, nc, ng, f, p) in ((1024, 1500, 300, 50, 0.2), (1056, 1200, 130, 7, 0.0), (1024, 1100, 257, 64, 0.1)):\n"
        "    g = t._graph(nc, ng, f, 4, dev, density=0.15, normalize_edges=False)\n"
        "    m = t._gsc_model(f, 20, 12, 'sum', p, 0.0, dev)\n"
        "    st = GraphSCStepper(m.model, g, b, torch.optim.Adam(m.model.parameters(), lr=1e-2, fused=True))\n"
        "    st.cfg.seed = 1234\n"
        "    seeds = (ng + torch.randperm(nc, generator=torch.Generator().manual_seed(1))[:b]).to(dev)\n"
        "    out.append(st.aggregate(seeds).cpu().numpy())\n"
        "np.savez(sys.argv[1], *out)\n")
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for mode in ("gather", "mfma"):
            path = os.path.join(tmp, mode + ".npz")
            env = dict(os.environ, DANCE_AMD_GRAPHSC_AGG=mode)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            res[mode] = dict(np.load(path))
    for k in res["gather"]:
        assert rel_err(res["mfma"][k], res["gather"][k]) < 1e-5, k
        assert not np.array_equal(res["mfma"][k], np.zeros_like(res["mfma"][k]))


def test_graphsc_fit_ministep_dropout_is_keyed_by_torch_seed(cuda_device):
    """With the reference's default dropouts (0.1 / 0.1) a fit is reproducible from ``torch.manual_seed`` (the Philox key is drawn from
    torch's generator), differs under another seed, and trains (finite losses that move)."""
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    g = _graph(600, 80, 50, 5, cuda_device, normalize_edges=False)
    out = []
    for seed in (11, 11, 12):
        torch.manual_seed(seed)
        m = GraphSC(in_feats=50, n_clusters=3, device="cuda")
        m.shuffle_generator = torch.Generator().manual_seed(1)
        torch.manual_seed(seed + 100)
        m.fit(g, epochs=2, lr=1e-3, batch_size=128)
        assert m.step_mode == "ministep"
        out.append((np.asarray(m.losses), m.get_latent().copy()))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert not np.array_equal(out[0][0], out[2][0])
    assert np.isfinite(out[0][0]).all() and np.isfinite(out[0][1]).all() and len(set(out[0][0].tolist())) > 1


# ---- scDeepSort -----------------------------------------------------------------------------------------------------------------------
def _sds_model(d, h, c, n_genes, p, dev, seed=0, compute_dtype="fp32"):
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import GNN
    torch.manual_seed(seed)
    m = GNN(d, c, h, 1, n_genes, activation=nn.ReLU(), dropout=p, compute_dtype=compute_dtype).to(dev)
    with torch.no_grad():
        m.alpha.uniform_(0.5, 1.5)
        m.layers[0].layers[1].bias.uniform_(-0.1, 0.1)
        m.linear.bias.uniform_(-0.1, 0.1)
    m.train()
    return m


def _sds_params(model):
    lin = model.layers[0].layers[1]
    return {"W1": lin.weight, "b1": lin.bias, "W2": model.linear.weight, "b2": model.linear.bias}


@pytest.mark.parametrize("b,n_genes,d,h,c,p,wd,bf16", [(64, 30, 32, 12, 5, 0.0, 0.0, False), (500, 120, 400, 200, 16, 0.0, 0.0, False),
                                                       (37, 20, 33, 7, 33, 0.2, 0.01, False), (96, 50, 64, 40, 3, 0.0, 0.0, True),
                                                       (130, 40, 136, 260, 48, 0.1, 0.0, False), (1, 10, 8, 4, 2, 0.0, 0.0, False),
                                                       (70, 25, 1000, 16, 64, 0.0, 0.0, True), (64, 30, 34, 12, 5, 0.0, 0.0, True)])
def test_scdeepsort_steps_vs_oracle(cuda_device, b, n_genes, d, h, c, p, wd, bf16):
    """Two consecutive steps of dh_scdeepsort_steps (gathered feature rows, fp32 and bf16 storage, dropout, weight decay, ragged tiles,
    up to 64 classes, vector and scalar aggregation kernels): losses, gradients (phase 1), updated parameters and the discarded aggregation against the float64 oracle."""
    from dance_amd import _lib
    from dance_amd.ministep import ScDeepSortStepper
    n_cells = max(3 * b, 64)
    g = _graph(n_cells, n_genes, d, 7, cuda_device)
    labels = torch.from_numpy(np.random.default_rng(1).integers(0, c, n_cells))
    g.ndata["label"] = torch.cat((-torch.ones(n_genes, dtype=torch.long), labels)).to(cuda_device)
    if bf16:
        g = g.with_ndata(features=g.ndata["features"].to(torch.bfloat16))
    rowptr, col, val = _host(g)
    feats = g.ndata["features"].float().cpu().numpy()
    m = _sds_model(d, h, c, n_genes, p, cuda_device)
    optim = torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=wd, fused=True)
    assert ScDeepSortStepper.eligible(m, g, b, optim, c)
    st = ScDeepSortStepper(m, g, b, optim)
    n_steps = 2
    seeds = (n_genes + torch.randperm(n_cells, generator=torch.Generator().manual_seed(3))[:n_steps * b]).to(cuda_device)
    lab_h = g.ndata["label"].cpu().numpy()
    # phase 1: gradients only
    st.grads = torch.zeros(sum(q.numel() for q in st.params), device=cuda_device)
    l1 = torch.empty(1, device=cuda_device)
    st.cfg.grads, st.cfg.phase, st.cfg.seeds, st.cfg.loss_out, st.cfg.dropout = st.grads.data_ptr(), 1, seeds.data_ptr(), l1.data_ptr(), p
    st._run(_lib.load().dh_scdeepsort_steps, "scdeepsort_steps", 0, 1)
    params = {k: v.detach().cpu().numpy().astype(np.float64) for k, v in _sds_params(m).items()}
    loss0, grads0 = oms.scdeepsort_step(params, feats, lab_h, seeds[:b].cpu().numpy(), dropout=p, seed=st.cfg.seed, step=0)
    flat, off = st.grads.cpu().numpy(), 0
    for k in ("W1", "b1", "W2", "b2"):
        got = flat[off:off + grads0[k].size].reshape(grads0[k].shape)
        off += grads0[k].size
        assert rel_err(got, grads0[k]) < 1e-4, k
    assert abs(float(l1) - loss0) < 1e-5 * abs(loss0)
    neigh_ref = oms.sage_neigh(rowptr, col, val, feats, g.ndata["cell_id"].cpu().numpy(), m.alpha.detach().cpu().numpy().ravel(), n_genes, seeds[:b].cpu().numpy())
    assert rel_err(st.neigh.cpu().numpy(), neigh_ref) < 1e-5
    for t_ in optim.state.values():
        t_["step"].zero_()
    st.cfg.phase, st.cfg.grads, st.grads = 0, None, None
    loss = torch.empty(n_steps, device=cuda_device)
    st.run(seeds, n_steps, loss)
    st.check_flags("test")
    mom = {k: np.zeros_like(v) for k, v in params.items()}
    var = {k: np.zeros_like(v) for k, v in params.items()}
    for s in range(n_steps):
        lo, grads = oms.scdeepsort_step(params, feats, lab_h, seeds[s * b:(s + 1) * b].cpu().numpy(), dropout=p, seed=st.cfg.seed, step=s)
        assert abs(float(loss[s]) - lo) < 2e-4 * abs(lo), s
        for k in params:
            pn, mn, vn = oms.adam_update(torch.from_numpy(params[k]), torch.from_numpy(grads[k]), torch.from_numpy(mom[k]), torch.from_numpy(var[k]),
                                         s + 1, 1e-2, 0.9, 0.999, 1e-8, wd)
            params[k], mom[k], var[k] = pn.numpy(), mn.numpy(), vn.numpy()
    for k, q in _sds_params(m).items():
        assert rel_err(q.detach().cpu().numpy(), params[k]) < 2e-3, k
    assert m.alpha.grad is None and m.layers[0].last_neigh is st.neigh


def test_scdeepsort_steps_flag_bad_labels(cuda_device):
    from dance_amd.ministep import ScDeepSortStepper
    g = _graph(64, 10, 8, 2, cuda_device)
    g.ndata["label"] = torch.cat((-torch.ones(10, dtype=torch.long), torch.full((64, ), 9))).to(cuda_device)
    m = _sds_model(8, 4, 3, 10, 0.0, cuda_device)
    optim = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
    st = ScDeepSortStepper(m, g, 8, optim)
    st.run(torch.arange(10, 18, device=cuda_device), 1, torch.empty(1, device=cuda_device))
    with pytest.raises(RuntimeError, match="label outside"):
        st.check_flags("test")


def test_scdeepsort_fit_ministep_equals_general_loop(cuda_device, tmp_path, monkeypatch):
    """ScDeepSort.fit on the persistent step ends where the general (eager) loop ends — same split and batch order from
    ``shuffle_generator`` — in fp32 and with bf16 feature storage; the step's two phases (gradients, then update: the data-parallel
    form) give bit-identical parameters to the fused step."""
    from dance_amd.modules.single_modality.cell_type_annotation import scdeepsort
    n_cells, n_genes, d = 700, 120, 32
    g = _graph(n_cells, n_genes, d, 3, cuda_device)
    labels = torch.from_numpy(np.random.default_rng(0).integers(0, 5, n_cells))
    for cd, tol in (("fp32", 1e-5), ("bf16", 2e-2)):
        out = {}
        for on in (True, False):
            monkeypatch.setattr(scdeepsort, "MINISTEP", on)
            monkeypatch.setattr(scdeepsort, "HIPGRAPH", False)
            torch.manual_seed(7)
            m = scdeepsort.ScDeepSort(d, 16, 1, "synthetic", f"mini{on}{cd}", batch_size=64, device="cuda", save_root=tmp_path, verbose=False, compute_dtype=cd)
            m.shuffle_generator = torch.Generator().manual_seed(11)
            m.fit(g, labels, epochs=3, lr=1e-2, val_ratio=0.2)
            assert m._use_mini == on and (m._stepper is not None) == on
            out[on] = ({k: v.detach().float().cpu().numpy() for k, v in m.model.state_dict().items()}, m.predict_proba(g))
        for k in out[True][0]:
            assert rel_err(out[True][0][k], out[False][0][k]) < tol, (cd, k)
        assert np.abs(out[True][1] - out[False][1]).max() < tol * 10


@pytest.mark.parametrize("which", ["graphsc", "scdeepsort"])
def test_data_parallel_phases_equal_the_fused_step(cuda_device, which):
    """The data-parallel form of a step — gradients into a flat buffer (phase 1), [the caller's all-reduce], Adam alone (phase 2) — ends
    on bit-identical parameters, moments and step counters as the fused step (phase 0): same gradient kernel, same update arithmetic."""
    from dance_amd import _lib
    from dance_amd.ministep import GraphSCStepper, ScDeepSortStepper
    lib = _lib.load()
    out = []
    for phases in (False, True):
        if which == "graphsc":
            g = _graph(300, 40, 10, 2, cuda_device, normalize_edges=False)
            m = _gsc_model(10, 12, 8, "sum", 0.1, 0.1, cuda_device).model
            optim = torch.optim.Adam(m.parameters(), lr=1e-2, fused=True)
            st = GraphSCStepper(m, g, 32, optim, world=2 if phases else 1)
            n_genes, fn, extra = 40, lib.dh_graphsc_steps, (torch.empty((64, 8), device=cuda_device), )
        else:
            g = _graph(300, 40, 16, 2, cuda_device)
            g.ndata["label"] = torch.cat((-torch.ones(40, dtype=torch.long), torch.arange(300) % 4)).to(cuda_device)
            m = _sds_model(16, 12, 4, 40, 0.1, cuda_device)
            optim = torch.optim.Adam(m.parameters(), lr=1e-2, weight_decay=0.01, fused=True)
            st = ScDeepSortStepper(m, g, 32, optim, world=2 if phases else 1)
            n_genes, fn, extra = 40, lib.dh_scdeepsort_steps, ()
        st.cfg.seed = 777
        seeds = (n_genes + torch.randperm(300, generator=torch.Generator().manual_seed(5))[:64]).to(cuda_device)
        loss = torch.empty(2, device=cuda_device)
        if not phases:
            st.run(seeds, 2, *extra, loss)
        else:  # what run() does with world > 1, minus the all-reduce (one process here)
            c = st.cfg
            c.seeds, c.loss_out = seeds.data_ptr(), loss.data_ptr()
            if extra:
                c.z_out = extra[0].data_ptr()
            c.dropout = 0.1
            for s in range(2):
                c.phase = 1
                st._run(fn, "steps", s, 1)
                c.phase = 2
                st._run(fn, "steps", s, 1)
        st.check_flags("test")
        out.append(([p.detach().clone() for p in st.params], [optim.state[p]["exp_avg_sq"].clone() for p in st.params],
                    [float(optim.state[p]["step"]) for p in st.params], loss.clone()))
    for a, b in zip(out[0][0] + out[0][1], out[1][0] + out[1][1]):
        assert torch.equal(a, b)
    assert out[0][2] == out[1][2] == [2.0] * 4 and torch.equal(out[0][3], out[1][3])
