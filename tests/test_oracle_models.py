"""oracle/models.py (the plain-torch CPU restatements bench.py times as ``cpu_baseline`` for BASELINE configs 2, 3, 5) against the
goldens the REFERENCE'S OWN classes produced: ScDSCModel.forward (model_heads.npz), SimpleGCDEC forward / target / loss / gradients
(model_heads.npz), the scDeepSort GNN + training batch (scdeepsort.npz: the first epoch of the reference's own fit).  CPU only."""
import json
import os

import numpy as np
import scipy.sparse as sp
import torch

from conftest import rel_err
from oracle import graphs as og
from oracle import models as om
from oracle.layers import scipy_to_torch_coo

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_scdsc_model_port_matches_reference_forward():
    h = np.load(os.path.join(GOLDEN, "model_heads.npz"))
    kw = json.loads(str(h["scdsc_kw"]))
    model = om.ScDSCModel(**kw).eval()
    sd = {k.split("::", 1)[1]: torch.from_numpy(h[k]) for k in h.files if k.startswith("scdsc_sd::")}
    assert sorted(sd) == sorted(model.state_dict())   # the reference's own parameter / buffer names
    model.load_state_dict(sd)
    n = h["scdsc_x"].shape[0]
    a = sp.csr_matrix((h["scdsc_adj_data"], h["scdsc_adj_indices"], h["scdsc_adj_indptr"]), shape=(n, n))
    with torch.no_grad():
        x_bar, q, predict, z3, mean, disp, pi = model(torch.from_numpy(h["scdsc_x"]), scipy_to_torch_coo(a))
    for name, val in (("x_bar", x_bar), ("q", q), ("predict", predict), ("z3", z3), ("mean", mean), ("disp", disp), ("pi", pi)):
        assert rel_err(val.numpy(), h[f"scdsc_{name}"]) < 1e-6, name


def test_scdsc_epoch_port_runs_and_descends():
    """scdsc_epoch (scdsc.py:270-287) is a well-formed training step: the loss is finite, only the un-frozen parameters move, and the
    ZINB term equals the float64 definition on the step's own tensors."""
    torch.manual_seed(0)
    n, g = 60, 20
    model = om.ScDSCModel(sigma=0.5, n_enc_1=24, n_enc_2=16, n_enc_3=16, n_dec_1=16, n_dec_2=16, n_dec_3=24, n_z1=16, n_z2=12, n_z3=8, n_clusters=4,
                          n_input=g)
    a = sp.random(n, n, density=0.1, random_state=1, format="csr", dtype=np.float32)
    a = (a + a.T + sp.eye(n)).tocsr()
    adj = scipy_to_torch_coo(sp.diags(1.0 / np.asarray(a.sum(1)).ravel()).dot(a).tocsr().astype(np.float32))   # row-normalised, as scdsc.py:130-132 feeds it
    data = torch.randn(n, g)
    x_raw = torch.poisson(torch.rand(n, g) * 3)
    sf = torch.rand(n, dtype=torch.float64) + 0.5
    with torch.no_grad():
        p = om.scdsc_target(model(data, adj)[1])
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    w0 = model.gnn_1.weight.detach().clone()
    losses = [om.scdsc_epoch(model, opt, data, adj, x_raw, sf, p) for _ in range(5)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert not torch.equal(w0, model.gnn_1.weight)


def test_gcdec_port_matches_reference():
    h = np.load(os.path.join(GOLDEN, "model_heads.npz"))
    d = h["gcdec_x"].shape[1]
    m = om.SimpleGCDEC(d, d)
    with torch.no_grad():
        m.gc.weight.copy_(torch.from_numpy(h["gcdec_w"]))
        m.gc.bias.copy_(torch.from_numpy(h["gcdec_b"]))
    m.mu = torch.nn.Parameter(torch.from_numpy(h["gcdec_mu"]))
    z, q = m(torch.from_numpy(h["gcdec_x"]), torch.from_numpy(h["gcdec_adj"]))
    p = m.target_distribution(q)
    loss = m.loss_function(p.data, q)
    loss.backward()
    assert rel_err(z.detach().numpy(), h["gcdec_z"]) < 1e-6 and rel_err(q.detach().numpy(), h["gcdec_q"]) < 1e-6
    assert rel_err(p.detach().numpy(), h["gcdec_p"]) < 1e-6 and abs(float(loss) - float(h["gcdec_loss"])) < 1e-6
    assert rel_err(m.gc.weight.grad.numpy(), h["gcdec_dw"]) < 1e-5 and rel_err(m.mu.grad.numpy(), h["gcdec_dmu"]) < 1e-5
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    labels, l2 = om.spagcn_iteration(m, opt, torch.from_numpy(h["gcdec_x"]), torch.from_numpy(h["gcdec_adj"]), p.data)
    assert labels.shape == (h["gcdec_x"].shape[0], ) and abs(l2 - float(h["gcdec_loss"])) < 1e-6


def test_scdeepsort_port_reproduces_the_reference_fit_first_epoch():
    """The port's block + GNN + summed-CE Adam step, driven with the golden's split and loader order, reproduces the first-epoch loss
    of the reference's own ``ScDeepSort.fit`` (case "mb": batches of 64) and the discarded ``neigh`` equals oracle.sage's."""
    gold = np.load(os.path.join(GOLDEN, "scdeepsort.npz"))
    kw = json.loads(str(gold["kw"]))
    case = kw["cases"]["mb"]
    x, labels = gold["x"], gold["labels"]
    n_cells, n_genes = x.shape
    e = og.cell_feature_graph(x, normalize_edges=True)
    order = np.argsort(e["dst"], kind="stable")
    n_nodes = n_genes + n_cells
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(e["dst"], minlength=n_nodes))
    col, val = e["src"][order], e["weight"][order].astype(np.float32)
    feats = torch.from_numpy(np.vstack((gold["gene_feat"], gold["cell_feat"])).astype(np.float32))
    cell_id = torch.from_numpy(e["cell_id"].astype(np.int64))
    model = om.ScDeepSortGNN(feats.shape[1], kw["hid"], int(labels.max()) + 1, n_genes)
    sd0 = {k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("mb_sd0::")}
    with torch.no_grad():
        model.alpha.copy_(sd0["alpha"])
        model.lin.weight.copy_(sd0["layers.0.layers.1.weight"])
        model.lin.bias.copy_(sd0["layers.0.layers.1.bias"])
        model.linear.weight.copy_(sd0["linear.weight"])
        model.linear.bias.copy_(sd0["linear.bias"])
    gen = torch.Generator().manual_seed(kw["seed_order"])
    perm = torch.randperm(n_cells, generator=gen) + n_genes                  # scdeepsort.py:157
    num_val = int(n_cells * 0.2)
    train_idx = perm[num_val:]
    full_labels = torch.cat((-torch.ones(n_genes, dtype=torch.long), torch.from_numpy(labels)))
    opt = torch.optim.Adam(model.parameters(), lr=case["lr"])
    order_ = train_idx[torch.randperm(train_idx.numel(), generator=gen)]     # the loader's shuffle
    tot = size = 0
    for i in range(0, order_.numel(), case["batch_size"]):
        seeds = order_[i:i + case["batch_size"]].numpy()
        loss = om.scdeepsort_batch(model, opt, rowptr, col, val, feats, cell_id, full_labels, seeds)
        tot, size = tot + loss * len(seeds), size + len(seeds)
    assert abs(tot / size - gold["mb_losses"][0]) < 2e-4 * gold["mb_losses"][0]


def test_graphsc_port_reproduces_the_reference_fit_losses():
    """The graph-sc port (block, WeightedGraphConv + Linear + inner-product decoder, the weighted BCE against the block's dst x dst
    adjacency, Adam) driven with the golden's loader order reproduces EVERY per-batch loss of the reference's own ``GraphSC.fit``
    (graphsc.npz, cases "mb": 3 epochs x 3 batches of 16, and "mean") and ends at the same weights."""
    gold = np.load(os.path.join(GOLDEN, "graphsc.npz"))
    kw = json.loads(str(gold["gsc_kw"]))
    x = gold["gsc_x"]
    n_cells, n_genes = x.shape
    e = og.cell_feature_graph(x, normalize_edges=False)
    order = np.argsort(e["dst"], kind="stable")
    n_nodes = n_genes + n_cells
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(e["dst"], minlength=n_nodes))
    col, val = e["src"][order], e["weight"][order].astype(np.float32)
    feats = torch.from_numpy(np.vstack((gold["gsc_gene_feat"], gold["gsc_cell_feat"])).astype(np.float32))
    for tag, batch, agg in (("mb", 16, "sum"), ("mean", 16, "mean")):
        model = om.GraphSCAE(kw["in_feats"], kw["hidden_dim"], (kw["hidden_1"], ), agg=agg, decoder_dropout=0.0)
        model.load_state_dict({k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f"gsc_{tag}_sd0::")})
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        gen = torch.Generator().manual_seed(123)
        train_ids = torch.arange(n_genes, n_nodes)
        losses, z = [], None
        for _ in range(3):
            ids = train_ids[torch.randperm(train_ids.numel(), generator=gen)]
            embs = []
            for i in range(0, ids.numel(), batch):
                loss, emb = om.graphsc_batch(model, opt, rowptr, col, val, feats, ids[i:i + batch].numpy())
                losses.append(loss)
                embs.append(emb)
            z = torch.cat(embs)[torch.argsort(ids)]   # graphsc.py:222-228: the epoch's embeddings in node order
        ref = gold[f"gsc_{tag}_losses"]
        assert len(losses) == len(ref) and np.allclose(losses, ref, rtol=2e-5, atol=0), (tag, losses, ref)
        assert np.allclose(z.numpy(), gold[f"gsc_{tag}_z"], rtol=1e-4, atol=1e-5)
        for k, v in model.state_dict().items():
            assert np.allclose(v.numpy(), gold[f"gsc_{tag}_sd1::{k}"], rtol=1e-4, atol=1e-6), (tag, k)


def test_scdsc_epoch_flop_count_of_the_bench_row():
    """scripts/bench_configs._scdsc_flops: what an epoch of the joint loop multiplies, per GEMM tag — the three trainable ZINB heads
    (forward = nt, dW = tn; no dX: their input comes from the frozen decoder), the seven GCN layers (X W = nn, dW = tn, dX of layers
    2..7 = nt) and NOT the frozen autoencoder (its outputs are computed once per fit).  A wrong count here mis-states the row's
    matrix-core fraction (it read 0.27 instead of 0.89 for `tn` until the heads were counted)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import bench_configs as bc
    n = 1000
    f = bc._scdsc_flops(n)
    heads = 3 * 512 * 2000
    gnn = 2000 * 512 + 512 * 256 + 256 * 256 + 256 * 256 + 256 * 128 + 128 * 32 + 32 * 10
    assert f["gemm_f32_nn"] == 2.0 * n * gnn
    assert f["gemm_f32_tn"] == 2.0 * n * (gnn + heads)
    assert f["gemm_f32_nt"] == 2.0 * n * (heads + gnn - 2000 * 512)
    # the model's own parameter shapes say the same
    m = om.ScDSCModel(sigma=0.5, n_clusters=10, n_input=2000)
    w = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert w["_dec_mean.0.weight"] == (2000, 512) and w["gnn_1.weight"] == (2000, 512) and w["gnn_7.weight"] == (32, 10)
