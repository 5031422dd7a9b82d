"""CPU: oracle/ministep.py (the checker of the persistent mini-batch steps, csrc/ministep.hip) pinned —
* its Philox4x32-10 against the Random123 known-answer vectors (kat_vectors of the Random123 distribution, philox4x32 with 10 rounds);
* its graph-sc step (dropout off) against EVERY per-batch loss and the final weights of the reference's own ``GraphSC.fit``
  (tests/golden/graphsc.npz, produced by the reference's classes: tests/golden/make_golden.py);
* its scDeepSort step against the first-epoch loss of the reference's own ``ScDeepSort.fit`` (tests/golden/scdeepsort.npz)."""
import json
import os

import numpy as np
import torch

from oracle import graphs as og
from oracle import ministep as oms

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_philox_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff, ) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = oms.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(x) for x in got) == want


def test_dropout_mask_rate_and_keying():
    m = oms.dropout_mask(200_000, 0.1, seed=1234, step=5, sid=0)
    assert set(np.unique(m)) == {0.0, float(np.float32(1.0 / (1.0 - float(np.float32(0.1)))))}
    assert abs((m > 0).mean() - 0.9) < 3e-3
    assert np.array_equal(m, oms.dropout_mask(200_000, 0.1, seed=1234, step=5, sid=0))
    for other in (dict(seed=1235, step=5, sid=0), dict(seed=1234, step=6, sid=0), dict(seed=1234, step=5, sid=1)):
        assert not np.array_equal(m, oms.dropout_mask(200_000, 0.1, **other))
    assert np.array_equal(oms.dropout_mask(7, 0.0, 1, 2, 3), np.ones(7))
    # a prefix of a longer draw is the shorter draw (element-keyed, not stream-keyed)
    assert np.array_equal(m[:1001], oms.dropout_mask(1001, 0.1, seed=1234, step=5, sid=0))


def _gsc_inputs(gold):
    x = gold["gsc_x"]
    n_cells, n_genes = x.shape
    e = og.cell_feature_graph(x, normalize_edges=False)
    order = np.argsort(e["dst"], kind="stable")
    n_nodes = n_genes + n_cells
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(e["dst"], minlength=n_nodes))
    col, val = e["src"][order], e["weight"][order].astype(np.float32)
    feats = np.vstack((gold["gsc_gene_feat"], gold["gsc_cell_feat"])).astype(np.float32)
    return rowptr, col, val, feats, n_cells, n_genes


def test_graphsc_step_reproduces_the_reference_fit():
    gold = np.load(os.path.join(GOLDEN, "graphsc.npz"))
    rowptr, col, val, feats, n_cells, n_genes = _gsc_inputs(gold)
    names = {"W1": "layer1.weight", "b1": "layer1.bias", "W2": "encoder.0.weight", "b2": "encoder.0.bias"}
    for tag, batch, agg in (("mb", 16, "sum"), ("mean", 16, "mean")):
        p = {k: gold[f"gsc_{tag}_sd0::{n}"].astype(np.float64) for k, n in names.items()}
        m = {k: np.zeros_like(v) for k, v in p.items()}
        v = {k: np.zeros_like(v_) for k, v_ in p.items()}
        gen = torch.Generator().manual_seed(123)
        ids_all = torch.arange(n_genes, n_genes + n_cells)
        losses, t = [], 0
        for _ in range(3):
            ids = ids_all[torch.randperm(ids_all.numel(), generator=gen)].numpy()
            for i in range(0, ids.size, batch):
                loss, _, grads = oms.graphsc_step(p, rowptr, col, val, feats, n_genes, ids[i:i + batch], agg=agg)
                t += 1
                for k in p:
                    pn, mn, vn = oms.adam_update(torch.from_numpy(p[k]), torch.from_numpy(grads[k]), torch.from_numpy(m[k]), torch.from_numpy(v[k]), t,
                                                 1e-2, 0.9, 0.999, 1e-8)
                    p[k], m[k], v[k] = pn.numpy(), mn.numpy(), vn.numpy()
                losses.append(loss)
        ref = gold[f"gsc_{tag}_losses"]
        assert len(losses) == len(ref) and np.allclose(losses, ref, rtol=2e-5, atol=0), (tag, losses, ref)
        for k, n in names.items():
            assert np.allclose(p[k], gold[f"gsc_{tag}_sd1::{n}"], rtol=1e-4, atol=1e-6), (tag, k)


def test_scdeepsort_step_reproduces_the_reference_first_epoch():
    gold = np.load(os.path.join(GOLDEN, "scdeepsort.npz"), allow_pickle=True)
    kw = json.loads(str(gold["kw"]))
    case = kw["cases"]["mb"]
    x, labels = gold["x"], gold["labels"]
    n_cells, n_genes = x.shape
    feats = np.vstack((gold["gene_feat"], gold["cell_feat"])).astype(np.float32)
    sd = {k.split("::", 1)[1]: gold[k] for k in gold.files if k.startswith("mb_sd0::")}
    p = {"W1": sd["layers.0.layers.1.weight"], "b1": sd["layers.0.layers.1.bias"], "W2": sd["linear.weight"], "b2": sd["linear.bias"]}
    p = {k: v.astype(np.float64) for k, v in p.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    vv = {k: np.zeros_like(v) for k, v in p.items()}
    gen = torch.Generator().manual_seed(kw["seed_order"])
    perm = torch.randperm(n_cells, generator=gen) + n_genes
    train_idx = perm[int(n_cells * 0.2):]
    order_ = train_idx[torch.randperm(train_idx.numel(), generator=gen)].numpy()
    full_labels = np.concatenate((-np.ones(n_genes, dtype=np.int64), labels.astype(np.int64)))
    tot = size = t = 0
    for i in range(0, order_.size, case["batch_size"]):
        seeds = order_[i:i + case["batch_size"]]
        loss, grads = oms.scdeepsort_step(p, feats, full_labels, seeds)
        t += 1
        for k in p:
            pn, mn, vn = oms.adam_update(torch.from_numpy(p[k]), torch.from_numpy(grads[k]), torch.from_numpy(m[k]), torch.from_numpy(vv[k]), t,
                                         case["lr"], 0.9, 0.999, 1e-8)
            p[k], m[k], vv[k] = pn.numpy(), mn.numpy(), vn.numpy()
        tot, size = tot + loss * len(seeds), size + len(seeds)   # scdeepsort.py:247-250 (the summed loss weighted by the batch size)
    assert abs(tot / size - gold["mb_losses"][0]) < 2e-4 * gold["mb_losses"][0]
