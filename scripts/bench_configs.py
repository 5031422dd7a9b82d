"""Model-level rows of the other BASELINE.json configs for bench.py's ``configs`` block (one GPU):

* ``c2_gcn_100k``            — the headline layer at config 2's size (100k cells x 2k genes, k = 15, fp32)
* ``c2_scdsc_epoch_100k/1M`` — one training epoch of ``ScDSC.fit`` (the headline layer's own model: autoencoder + 7 chained GCN layers
                               + ZINB, scdsc.py:253-288), timed as fit(E2 epochs) - fit(E1 epochs) of the product's own method
* ``c3_scdeepsort_1M_bf16_epoch`` — ``ScDeepSort.fit`` epoch (training pass + the reference's two evaluation passes) on the 1M-cell
                               cell-gene graph, bf16 storage, and the same in fp32
* ``c5_spagcn_500k_iter``    — one DEC iteration of SpaGCN's ``SimpleGCDEC.fit_with_init`` (spagcn.py:541-584) on 500k spots, spatial
                               kNN k = 15 truncated Gaussian adjacency, 50 -> 50

Each entry: ``ms`` (per epoch / iteration / step), ``kernels_ms`` (HIP-event time per C-ABI tag per unit), ``roofline`` of the dominant
kernel (algorithmic flops or bytes stated in ``basis``), and ``cpu_baseline`` = the plain-torch restatement (oracle/models.py,
oracle/layers.py; ``kind: "port"`` — the AST-lifted reference classes cannot travel to the GPU box, and DGL is not installable) timed on
a bounded sample drawn by the SAME generators as the GPU leg.  Inputs are generated on the device and resident before anything is
timed.  Every config is independent and guarded: a failure is reported in place and never costs the headline line.
"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS, PEAK_F32_TF, PEAK_BF16_TF = 8000.0, 157.3, 2500.0


def _kernel_totals(timer):
    return {k: v[0] * v[1] for k, v in timer.summary().items()}


def _per_unit(t2, t1, units):
    return {k: round((t2.get(k, 0.0) - t1.get(k, 0.0)) / units, 4) for k in sorted(set(t1) | set(t2)) if (t2.get(k, 0.0) - t1.get(k, 0.0)) / units > 5e-4}


def _cpu_time(fn, min_seconds=4.0, max_iters=5):
    fn()  # warm-up
    times, t_all = [], time.perf_counter()
    while len(times) < max_iters and (time.perf_counter() - t_all < min_seconds or len(times) < 2):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), len(times)


# ---- config 2: the layer at 100k cells -------------------------------------------------------------------------------------------
def c2_gcn_100k(dev, steps=20):
    import bench
    from dance_amd import autograd, kernels
    from dance_amd.graph import CSRGraph
    from oracle import layers as ol
    n, F_, H, K = 100_000, bench.N_GENES, bench.N_HIDDEN, bench.K_NEIGH
    rowptr, col, val = bench.synth_rand_graph(n, K, dev, seed=1)
    graph = CSRGraph(rowptr, col, val, n, n)
    graph.transpose()
    x = bench.synth_features(n, F_, dev, seed=100)
    bound = (6.0 / (F_ + H))**0.5
    w = ((torch.rand((F_, H), device=dev, generator=torch.Generator(device=dev).manual_seed(2)) * 2 - 1) * bound).requires_grad_(True)
    dy = torch.randn((n, H), device=dev, generator=torch.Generator(device=dev).manual_seed(3))

    def step():
        w.grad = None
        autograd.gcn_layer(x, w, graph, None, True).backward(dy)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with kernels.KernelTimer() as timer:
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    ks = {k: round(v[1], 4) for k, v in sorted(timer.summary().items())}
    dom = max(ks, key=ks.get)
    flops = 2.0 * n * F_ * H
    roof = {"kernel": dom, "bound": "mfma", "achieved": round(flops / ks[dom] / 1e9, 2), "peak": PEAK_F32_TF, "unit": "TFLOP/s",
            "frac": round(flops / ks[dom] / 1e9 / PEAK_F32_TF, 4), "basis": "2 N F H flops per GEMM launch",
            "layer_hbm_frac": round(bench.layer_bytes(n, K * n) / (ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4)}
    # CPU: the same tensors through the torch-CPU restatement of the reference layer
    xc, wc, dyc = x.cpu(), w.detach().cpu(), dy.cpu()
    rows = torch.arange(n).repeat_interleave(K)
    adj = torch.sparse_coo_tensor(torch.stack([rows, col.cpu().long()]), val.cpu(), (n, n))
    layer = ol.GNNLayer(F_, H)
    with torch.no_grad():
        layer.weight.copy_(wc)

    def cpu_step():
        layer.weight.grad = None
        layer(xc, adj).backward(dyc)
    med, it = _cpu_time(cpu_step)
    return {"workload": f"GCN layer (scDSC GNNLayer) fwd+bwd, {n} cells x {F_} genes -> {H}, rand-k{K}, fp32 (BASELINE config 2 size)",
            "ms": round(ms, 4), "value": n / (ms * 1e-3), "unit": "cells/s", "kernels_ms": ks, "roofline": roof,
            "cpu_baseline": {"value": n / med, "unit": "cells/s", "cores": torch.get_num_threads(), "kind": "port",
                             "sample": f"the GPU leg's own X, W, dY and graph, all {n} cells, oracle.layers.GNNLayer on torch-CPU, median of {it} ({med * 1e3:.0f} ms each)"}}


# ---- config 2's model: ScDSC.fit epochs --------------------------------------------------------------------------------------------
def _scdsc_inputs(n, dev, seed=11):
    """Standardised expression X (the headline generator), matching raw counts for the ZINB term, size factors, rand-k15 row-normalised graph."""
    import bench
    from dance_amd.graph import CSRGraph
    g = torch.Generator(device=dev).manual_seed(seed)
    x = bench.synth_features(n, bench.N_GENES, dev, seed=seed)
    counts = torch.empty((n, bench.N_GENES), dtype=torch.float32, device=dev)
    lam = torch.exp(torch.randn(bench.N_GENES, device=dev, generator=g))
    for lo in range(0, n, 125_000):
        hi = min(n, lo + 125_000)
        counts[lo:hi] = torch.poisson(lam[None, :].expand(hi - lo, -1).contiguous(), generator=g) * (torch.rand((hi - lo, bench.N_GENES), device=dev, generator=g) < 0.10)
    n_counts = counts.sum(1).clamp_(min=1.0)
    rowptr, col, val = bench.synth_rand_graph(n, bench.K_NEIGH, dev, seed=1)
    y = torch.randint(0, 10, (n, ), generator=torch.Generator().manual_seed(seed)).numpy()
    return x, counts, n_counts, CSRGraph(rowptr, col, val, n, n), y


def _scdsc_flops(n, g=2000, e1=512, e2=256, e3=256, z1=256, z2=128, z3=32, c=10):
    """Flops of ONE joint-training epoch per GEMM tag.  The autoencoder (incl. x_bar) is frozen and its outputs are kept for the whole
    fit (AE.cache_frozen), so it is not in an epoch; the three ZINB heads (dec_3 = 512 -> g) train: forward (nt) and dW (tn), no dX
    (their input is the frozen decoder's output).  GCN layers: X W (nn), dW (tn), dX = dS W^T of layers 2..7 (nt)."""
    heads = 3.0 * e1 * g
    gnn = g * e1 + e1 * e2 + e2 * e3 + e3 * z1 + z1 * z2 + z2 * z3 + z3 * c
    gnn_dx = e1 * e2 + e2 * e3 + e3 * z1 + z1 * z2 + z2 * z3 + z3 * c
    return {"gemm_f32_nt": 2.0 * n * (heads + gnn_dx), "gemm_f32_nn": 2.0 * n * gnn, "gemm_f32_tn": 2.0 * n * (gnn + heads)}


def c2_scdsc_epoch(dev, n, e1=1, e2=6, cpu_sample=8_000, cpu_baseline=None):
    from dance_amd import kernels
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    from oracle import models as om
    x, counts, n_counts, graph, y = _scdsc_inputs(n, dev)
    graph.transpose()
    xh, ch, nh = x.cpu().numpy(), counts.cpu().numpy(), n_counts.cpu().numpy().astype(np.float64)
    del x, counts
    torch.manual_seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        m = ScDSC(pretrain_path=os.path.join(tmp, "ae.pt"), sigma=0.5, n_clusters=10, n_input=xh.shape[1], device="cuda")

        def fit(epochs):
            torch.cuda.synchronize()
            with kernels.KernelTimer() as timer:
                t0 = time.perf_counter()
                m.fit((graph, xh, ch, nh), y, lr=1e-3, epochs=epochs, pt_epochs=0)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            return dt, _kernel_totals(timer)
        fit(1)  # warm-up: allocator, lazy kernels, the CSR transpose cache
        t_a, k_a = fit(e1)
        m.record_epoch_times = True
        t_b, k_b = fit(e2)
        # device time between the epoch boundaries of the longer fit, epoch 0 (which also runs the evaluation pass of every tenth epoch)
        # left out.  (Until round 6 this was (fit(e2) - fit(e1)) / (e2 - e1) by the wall clock: each fit copies 16 GB from pageable host
        # memory, whose +-5 % moved a three-epoch difference by +-15 ms per epoch — rows between 158 and 198 ms for the same kernels.)
        per_epoch = sorted(m.epoch_ms[1:])
    units = e2 - e1
    ms = per_epoch[len(per_epoch) // 2]
    ks = _per_unit(k_b, k_a, units)
    known = _scdsc_flops(n, xh.shape[1])
    dom = max(ks, key=ks.get)
    roof = {"kernel": dom, "ms": ks[dom]}
    if dom in known:
        roof.update(bound="mfma", achieved=round(known[dom] / ks[dom] / 1e9, 2), peak=PEAK_F32_TF, unit="TFLOP/s", frac=round(known[dom] / ks[dom] / 1e9 / PEAK_F32_TF, 4),
                    basis="sum of 2 M K N over the GEMMs of this tag in one epoch: nt = the three ZINB heads' forward + the GCN layers' dX; nn = GCN X W; "
                          "tn = GCN dW + the heads' dW (the frozen autoencoder is computed once per fit, not per epoch)",
                    all_tags={t: {"ms": ks[t], "TFLOP/s": round(known[t] / ks[t] / 1e9, 1), "frac": round(known[t] / ks[t] / 1e9 / PEAK_F32_TF, 3)} for t in known if t in ks})
    out = {"workload": f"ScDSC.fit, one joint-training epoch (full batch): AE 2000-512-256-256-[256-128-32]-256-256-512-2000 (frozen: computed once per fit) "
                       f"+ 7 GCN layers + 3 ZINB heads + ZINB loss, {n} cells x 2000 genes, rand-k15, fp32; median device time of epochs 1..{e2 - 1} of the product's own fit({e2}) "
                       f"(kernels_ms: fit({e2}) - fit({e1}))",
           "ms": round(ms, 3), "value": n / (ms * 1e-3), "unit": "cells/s per epoch", "kernels_ms": ks, "other_ms": round(ms - sum(ks.values()), 3), "roofline": roof}
    if cpu_baseline is not None:  # the per-cell rate of the same port, measured once (run_all)
        out["cpu_baseline"] = cpu_baseline
        return out
    # CPU: the restated model + training step on a sample of the same generators
    ns = min(cpu_sample, n)
    import bench
    xs, cs, ncs, gs, _ = _scdsc_inputs(ns, torch.device("cpu"))
    rows = torch.arange(ns).repeat_interleave(bench.K_NEIGH)
    adj = torch.sparse_coo_tensor(torch.stack([rows, gs.col.long()]), gs.val, (ns, ns))
    torch.manual_seed(0)
    ref = om.ScDSCModel(sigma=0.5, n_clusters=10, n_input=xs.shape[1])
    for p_ in ref.ae.parameters():   # fix_module("model.ae") (scdsc.py:110): the autoencoder is frozen in the joint loop
        p_.requires_grad_(False)
    opt = torch.optim.Adam([p_ for p_ in ref.parameters() if p_.requires_grad], lr=1e-3)
    sf = (ncs.double() / ncs.double().median())
    with torch.no_grad():
        p_t = om.scdsc_target(ref(xs, adj)[1])
    med, it = _cpu_time(lambda: om.scdsc_epoch(ref, opt, xs, adj, cs, sf, p_t), min_seconds=3.0, max_iters=2)
    out["cpu_baseline"] = {"value": ns / med, "unit": "cells/s per epoch", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"{ns} cells drawn by the same generators, oracle.models.ScDSCModel + scdsc_epoch on torch-CPU (the whole model every "
                                     f"epoch, as the reference runs it), median of {it} ({med * 1e3:.0f} ms each)"}
    return out


# ---- config 3: ScDeepSort.fit epoch on the 1M-cell cell-gene graph ----------------------------------------------------------------
def _cellgene_graph(n_cells, n_genes, per, dfeat, dev, seed=0):
    from dance_amd import kernels
    from dance_amd.cellgraph import CellGeneGraph
    g = torch.Generator(device=dev).manual_seed(seed)
    col = torch.empty((n_cells, per), dtype=torch.int32, device=dev)
    for lo in range(0, n_cells, 125_000):
        hi = min(n_cells, lo + 125_000)
        col[lo:hi] = torch.rand(hi - lo, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32)
    col = col.reshape(-1)
    rp_x = torch.arange(0, n_cells * per + 1, per, dtype=torch.int32, device=dev)
    val_x = torch.rand(n_cells * per, device=dev, generator=g) + 0.5
    rp_t, col_t, val_t, perm_t = kernels.csr_transpose(rp_x, col, val_x, n_cells, n_genes)
    vx, vt = kernels.csr_row_normalize(rp_x, val_x), kernels.csr_row_normalize(rp_t, val_t)
    rowptr, gcol, gval, eid = kernels.cellgene_graph_assemble(rp_x, col, vx, rp_t, col_t, vt, perm_t, n_cells, n_genes)
    n_nodes = n_genes + n_cells
    feats = torch.randn(n_nodes, dfeat, device=dev, generator=g)
    cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
    fid = torch.cat((-torch.ones(n_genes, dtype=torch.int32), torch.arange(n_cells, dtype=torch.int32))).to(dev)
    return CellGeneGraph(rowptr, gcol, gval, eid, n_nodes, {"cell_id": cid, "feat_id": fid, "features": feats})


def c3_scdeepsort_epoch(dev, n_cells=1_000_000, batch=65536, cpu_cells=20_000, ref_batch=500):
    from dance_amd import kernels
    from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import ScDeepSort
    from oracle import models as om
    n_genes, dfeat, hid, per = 2000, 400, 200, 200
    cg = _cellgene_graph(n_cells, n_genes, per, dfeat, dev)
    labels = torch.randint(0, 16, (n_cells, ), generator=torch.Generator().manual_seed(0))
    out = {}
    for cd in ("fp32", "bf16"):
        with tempfile.TemporaryDirectory() as tmp:
            m = ScDeepSort(dfeat, hid, 1, "synthetic", "c3", batch_size=batch, device="cuda", save_root=tmp, verbose=False, compute_dtype=cd)
            torch.manual_seed(0)
            m.fit(cg, labels, epochs=1, lr=1e-3, val_ratio=0.2)  # warm-up
            steady = None
            if cd == "bf16":  # the steady-state epoch, as configs 2 and 5 measure theirs: fit(4) - fit(1) over 3 (a fit call has fixed costs)
                tt = {}
                for e in (1, 7, 1, 7):  # best of two each: a fit call writes a checkpoint whenever validation accuracy improves (~2 ms)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    m.fit(cg, labels, epochs=e, lr=1e-3, val_ratio=0.2)
                    torch.cuda.synchronize()
                    tt[e] = min(tt.get(e, 1e9), time.perf_counter() - t0)
                steady_wall = (tt[7] - tt[1]) / 6
                if steady_wall <= 0:  # toy sizes: the call's fixed costs (and their jitter) exceed six epochs; an upper bound then
                    steady_wall = tt[7] / 7
                # device time between the epoch boundaries of one more fit(7), first epoch left out, median: the wall-clock difference of two
                # fits above gave 15.8 - 25.4 ms for the same kernels (a fit call's fixed costs move by more than six epochs take)
                m.record_epoch_times = True
                m.fit(cg, labels, epochs=7, lr=1e-3, val_ratio=0.2)
                m.record_epoch_times = False
                ev = sorted(m.epoch_ms[1:])
                steady = ev[len(ev) // 2] * 1e-3
                out["steady_wall"] = steady_wall
            best = None
            for _ in range(2):  # best of two: a fit call also writes a checkpoint
                torch.cuda.synchronize()
                with kernels.KernelTimer() as timer:
                    t0 = time.perf_counter()
                    m.fit(cg, labels, epochs=1, lr=1e-3, val_ratio=0.2)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, {k: round(v, 3) for k, v in _kernel_totals(timer).items()})
        out[cd] = best
        if steady is not None:
            out["steady"] = steady
    # the reference's default batch (scdeepsort.py:115): every full training batch of an epoch behind one C call (csrc/ministep.hip)
    ref_rows = {}
    for cd in ("fp32", "bf16"):
        with tempfile.TemporaryDirectory() as tmp:
            m = ScDeepSort(dfeat, hid, 1, "synthetic", "c3ref", batch_size=ref_batch, device="cuda", save_root=tmp, verbose=False, compute_dtype=cd)
            torch.manual_seed(0)
            m.fit(cg, labels, epochs=1, lr=1e-3, val_ratio=0.2)
            tt = {}
            for e in (1, 3, 1, 3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                m.fit(cg, labels, epochs=e, lr=1e-3, val_ratio=0.2)
                torch.cuda.synchronize()
                tt[e] = min(tt.get(e, 1e9), time.perf_counter() - t0)
            ep = (tt[3] - tt[1]) / 2 if tt[3] > tt[1] else tt[3] / 3
            # the training steps alone: HIP events around one C call over the epoch's full batches
            st, graph = m._stepper, m._typed(cg.to("cuda"))
            step_ms = None
            if st is not None:
                n_steps = min(400, int(n_cells * 0.8) // ref_batch)
                seeds = (n_genes + torch.randperm(n_cells, device=dev))[:n_steps * ref_batch].contiguous()
                loss = torch.empty(n_steps, device=dev)
                st.run(seeds, n_steps, loss)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                st.run(seeds, n_steps, loss)
                b.record()
                torch.cuda.synchronize()
                step_ms = a.elapsed_time(b) / n_steps
            n_train_steps = int(n_cells * 0.8) // ref_batch
            ref_rows[cd] = {"ms": round(ep * 1e3, 1), "value": n_cells / ep, "train_steps": n_train_steps, "mode": "ministep" if m._use_mini else "hipgraph/eager",
                            "ms_per_train_step_incl_eval": round(ep * 1e3 / n_train_steps, 4), "ms_per_step": None if step_ms is None else round(step_ms, 4)}
    dt, ks = out["bf16"]
    dom = max(ks, key=ks.get)
    # one cell<-gene aggregation over the whole graph = nnz (4 + s) + 4 (N + 1) + G D s + N D s bytes
    nnz = n_cells * (per + 1)
    agg_bytes = nnz * (4 + 2) + 4.0 * (n_cells + 1) + n_genes * dfeat * 2 + n_cells * dfeat * 2
    # both bounds for the aggregation kernels of one epoch: they process 1.8 x the graph (training pass over 80 % of the cells + the
    # full-graph evaluation pass).  HBM: SURVEY 8(d)'s algorithmic bytes; MFMA: the dense-equivalent product the kernel actually runs
    # (cells x genes x D, x 2 split planes for bf16 storage: weights as bf16 hi + lo) against the dense bf16 peak
    sage_ms = sum(v for k_, v in ks.items() if k_.startswith("sage_window_mfma"))
    passes = 1.8
    hbm_gbs = agg_bytes * passes / (sage_ms * 1e-3) / 1e9 if sage_ms else None
    dense_tflop = 2.0 * n_cells * n_genes * dfeat * 2 * passes / 1e12
    roof = {"kernel": dom, "ms_per_epoch": ks[dom], "bound": "hbm", "achieved": None if hbm_gbs is None else round(hbm_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": None if hbm_gbs is None else round(hbm_gbs / PEAK_HBM_GBS, 4),
            "basis": f"SURVEY 8(d) bytes of one aggregation over the whole graph ({round(agg_bytes / 1e9, 3)} GB, bf16 storage) x {passes} passes per epoch "
                     f"/ the epoch's sage_window_mfma* kernel time ({round(sage_ms, 3)} ms)",
            "mfma": {"bound": "mfma bf16 (dense-equivalent product: 10x the useful flops at 10 % density, x 2 split planes)",
                     "achieved": None if not sage_ms else round(dense_tflop / (sage_ms * 1e-3), 1), "peak": PEAK_BF16_TF, "unit": "TFLOP/s",
                     "frac": None if not sage_ms else round(dense_tflop / (sage_ms * 1e-3) / PEAK_BF16_TF, 4)},
            "note": "the aggregation runs twice per epoch (training pass over 80 % of the cells in batches + one full-graph evaluation pass)",
            "aggregation_algorithmic_GB": round(agg_bytes / 1e9, 3)}
    # CPU: restated block path on a sample graph of the same generator, batches of 500 (the reference default)
    small = _cellgene_graph(cpu_cells, n_genes, per, dfeat, dev, seed=1)  # (the builders are HIP kernels: built on the GPU, moved to the host)
    rowptr, col, val = small.rowptr.cpu().numpy().astype(np.int64), small.col.cpu().numpy().astype(np.int64), small.val.cpu().numpy()
    feats, cid = small.ndata["features"].cpu(), small.ndata["cell_id"].cpu().long()
    full_labels = torch.cat((-torch.ones(n_genes, dtype=torch.long), labels[:cpu_cells]))
    ref = om.ScDeepSortGNN(dfeat, hid, 16, n_genes)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    seeds_all = np.random.default_rng(0).permutation(cpu_cells) + n_genes
    state = {"i": 0}

    def cpu_batch():
        i = state["i"]
        om.scdeepsort_batch(ref, opt, rowptr, col, val, feats, cid, full_labels, seeds_all[i:i + 500])
        state["i"] = (i + 500) % (cpu_cells - 500)
    med, it = _cpu_time(cpu_batch, min_seconds=6.0, max_iters=12)
    return {"workload": f"ScDeepSort.fit, one epoch = training pass (80 % of the cells, batch {batch}) + the reference's two evaluation passes, "
                        f"{n_cells} cells x {n_genes} genes at 10 % density (nnz {n_cells * per}), D = {dfeat} -> {hid}, bf16 storage + bf16 MFMA dense update",
            "ms": round(out["steady"] * 1e3, 2), "value": n_cells / out["steady"], "unit": "cells/s per epoch",
            "wall_clock_difference_ms": round(out.get("steady_wall", 0.0) * 1e3, 2),
            "ms_basis": "steady-state epoch = median device time between the epoch boundaries of a fit(7), first epoch left out (ScDeepSort.record_epoch_times; "
                        "wall_clock_difference_ms = (fit(7) - fit(1)) / 6, best of two calls each: what the row quoted before); fit_call_1_epoch_ms is a whole one-epoch fit call "
                        "(model construction, split, checkpoint included: the number rounds 3-4 quoted), kernels_ms belongs to that call",
            "fit_call_1_epoch_ms": round(dt * 1e3, 2), "kernels_ms": ks, "roofline": roof,
            "fp32": {"ms": round(out["fp32"][0] * 1e3, 2), "kernels_ms": out["fp32"][1]},
            "reference_batch": {"batch": ref_batch, **ref_rows["fp32"], "bf16_storage": ref_rows["bf16"],
                                "note": "the reference's default batch size and arithmetic (fp32): steady-state epoch = (fit(3) - fit(1)) / 2 — the training pass as ONE C call "
                                        "over all full batches (dh_scdeepsort_steps: discarded aggregation, gathered-row Linear on the fp32 matrix cores, cross entropy, "
                                        "weight gradients + Adam; 4 launches per step) + the two evaluation passes; ms_per_step = HIP events around the C call / steps"},
            "cpu_baseline": {"value": 500 / med, "unit": "training cells/s (one pass; an epoch is ~2.2 passes)", "cores": torch.get_num_threads(), "kind": "port",
                             "sample": f"graph of {cpu_cells} cells from the same generator, batches of 500 (reference default), oracle.models.scdeepsort_batch "
                                       f"(restated AdaptiveSAGE block path, fp32; DGL not installable), median of {it} batches ({med * 1e3:.0f} ms each)"}}


# ---- config 4 on one GPU: GraphSC.fit epoch on the 1M-cell cell-gene graph ----------------------------------------------------------
def c4_graphsc_epoch(dev, n_cells=1_000_000, batch=8192, ref_batch=128, cpu_cells=20_000, ref_batch_epochs=True):
    from dance_amd import kernels
    from dance_amd.modules.single_modality.clustering.graphsc import GraphSC
    from oracle import models as om
    n_genes, per, dfeat = 2000, 200, 50
    cg = _cellgene_graph(n_cells, n_genes, per, dfeat, dev)

    gs_ref = [None]

    def steady(bsz, e2, reps=1):
        torch.manual_seed(0)
        gs = GraphSC(in_feats=dfeat, n_clusters=10, device="cuda")
        gs_ref[0] = gs
        gs.fit(cg, epochs=1, batch_size=bsz)  # warm-up (allocator, capture below batch 2048)
        res = {}
        for e in (1, e2) * reps:  # best of `reps` calls each
            torch.cuda.synchronize()
            with kernels.KernelTimer() as timer:
                t0 = time.perf_counter()
                gs.fit(cg, epochs=e, batch_size=bsz)
                torch.cuda.synchronize()
                dt_ = time.perf_counter() - t0
            if e not in res or dt_ < res[e][0]:
                res[e] = (dt_, _kernel_totals(timer))
        # device time between epoch boundaries of one more fit (the last epoch, which also moves the embedding to the host, left out):
        # the wall-clock difference of two fits carries the run-to-run spread of their fixed costs (141 - 167 ms for the same kernels)
        gs.record_epoch_times = True
        gs.fit(cg, epochs=e2 + 1, batch_size=bsz)
        gs.record_epoch_times = False
        ev = sorted(gs.epoch_ms[1:-1])
        return ev[len(ev) // 2] * 1e-3, _per_unit(res[e2][1], res[1][1], e2 - 1), (res[e2][0] - res[1][0]) / (e2 - 1)
    dt, ks, dt_wall = steady(batch, 3, reps=2)
    out = {"workload": f"GraphSC.fit (graph-sc GAE: WeightedGraphConv {dfeat} -> 200, Linear 200 -> 300, inner-product decoder, weighted BCE on the block's "
                       f"dst x dst adjacency, two forwards per batch as graphsc.py:202,215 writes it), one epoch over {n_cells} cells x {n_genes} genes at 10 % "
                       f"density, batch {batch}, fp32, ONE GPU (BASELINE's config shards it over 8); steady-state epoch = median device time between the epoch "
                       f"boundaries of a fit(4), first and last epoch left out (kernels_ms: (fit(3) - fit(1)) / 2)",
           "ms": round(dt * 1e3, 2), "value": n_cells / dt, "unit": "cells/s per epoch", "wall_clock_difference_ms": round(dt_wall * 1e3, 2), "kernels_ms": ks}
    dom = max(ks, key=ks.get) if ks else None
    # the dominant kernel of the large-batch epoch is the all-pairs decoder (dh_gram_sigmoid_f32): ONE Gram per batch — the second forward's
    # (graphsc.py:215; the first forward's logits are never formed, :202-203) — i.e. two B x B x E products per batch (x = z z^T, O = sigmoid(x) z)
    n_b = n_cells // batch
    gram_flops = n_b * 2 * 2.0 * batch * batch * 300
    gm = ks.get("gram_sigmoid_f32")
    out["roofline"] = {"kernel": dom, "ms_per_epoch": ks.get(dom) if dom else None, "bound": "mfma",
                       "achieved": None if not gm else round(gram_flops / gm / 1e9, 2), "peak": PEAK_F32_TF, "unit": "TFLOP/s",
                       "frac": None if not gm else round(gram_flops / gm / 1e9 / PEAK_F32_TF, 4),
                       "basis": f"gram_sigmoid_f32: {n_b} batches x 2 products of 2 B^2 E flops (B = {batch}, E = 300; one Gram per batch: only the second forward has a decoder) "
                                f"/ its time per epoch, against the fp32 matrix-core peak",
                       "note": "step mode 'aggfirst': aggregation straight off the CSR rows (dh_graphsc_steps phase 3), dense layers / decoder / Adam on the big-tile kernels"}
    if ref_batch_epochs:
        dt_r, _, _ = steady(ref_batch, 2)
        out["reference_batch"] = {"batch": ref_batch, "ms": round(dt_r * 1e3, 1), "value": n_cells / dt_r, "ms_per_step": round(dt_r * 1e3 / -(-n_cells // ref_batch), 4),
                                  "mode": getattr(gs_ref[0], "step_mode", None),
                                  "note": "the reference's default batch size: all full batches of an epoch behind ONE C call (dh_graphsc_steps, csrc/ministep.hip: 4 launches per step, "
                                          "no block, no transposed copy, Adam in the gradient kernel); steady-state epoch = device time of the middle epoch of a fit(3), ms_per_step = that / steps"}
    # CPU: the restated loop at the reference's batch size on a sample graph of the same generator
    small = _cellgene_graph(cpu_cells, n_genes, per, dfeat, dev, seed=1)
    rowptr, col, val = small.rowptr.cpu().numpy().astype(np.int64), small.col.cpu().numpy().astype(np.int64), small.val.cpu().numpy()
    feats = small.ndata["features"].cpu()
    torch.manual_seed(0)
    ref = om.GraphSCAE(dfeat, 200, (300, ), decoder_dropout=0.1, dropout=0.1)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-5)
    seeds_all = np.random.default_rng(0).permutation(cpu_cells) + n_genes
    state = {"i": 0}

    def cpu_batch():
        i = state["i"]
        om.graphsc_batch(ref, opt, rowptr, col, val, feats, seeds_all[i:i + ref_batch])
        state["i"] = (i + ref_batch) % (cpu_cells - ref_batch)
    med, it = _cpu_time(cpu_batch, min_seconds=5.0, max_iters=40)
    out["cpu_baseline"] = {"value": ref_batch / med, "unit": "cells/s per epoch", "cores": torch.get_num_threads(), "kind": "port",
                           "sample": f"graph of {cpu_cells} cells from the same generator, batches of {ref_batch} (the reference default), oracle.models.graphsc_batch "
                                     f"(restated block + GCNAE + loss, pinned to every per-batch loss of the reference's own fit: tests/test_oracle_models.py; "
                                     f"DGL not installable), median of {it} batches ({med * 1e3:.1f} ms each)"}
    return out


# ---- config 5: SpaGCN DEC iteration at 500k spots ---------------------------------------------------------------------------------
def _spatial_graph(n, k, dev, seed=5):
    from dance_amd import kernels
    from dance_amd.graph import CSRGraph
    rng = np.random.default_rng(seed)
    side = int(np.ceil(np.sqrt(n)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = np.stack([gx.ravel() + 0.5 * (gy.ravel() % 2), gy.ravel() * 0.866], 1)[:n] + rng.normal(0, 0.05, (n, 2))
    xyz = torch.from_numpy(np.hstack([xy, rng.normal(0, 0.3, (n, 1))]).astype(np.float32)).to(dev)
    t0 = time.perf_counter()
    idx, dist = kernels.knn(xyz, k)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    order = torch.argsort(idx, dim=1)
    g = CSRGraph(torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev), torch.gather(idx, 1, order).reshape(-1).contiguous(),
                 torch.gather(dist, 1, order).reshape(-1).contiguous(), n, n)
    return g, build_s


def c5_spagcn_iter(dev, n=500_000, k=15, e1=3, e2=33, cpu_spots=50_000):
    from dance_amd import kernels
    from dance_amd.modules.spatial.spatial_domain.spagcn import SimpleGCDEC, SpaGCN
    from oracle import models as om
    g, build_s = _spatial_graph(n, k, dev)
    adj = SpaGCN(l=1.2, device=dev).calc_adj_exp(g)
    adj.transpose()
    gen = torch.Generator(device=dev).manual_seed(0)
    emb = torch.randn(n, 50, device=dev, generator=gen)
    init_y = np.random.default_rng(0).integers(0, 10, n)

    last_model = [None]

    def fit(epochs, record=False):
        torch.manual_seed(0)
        m = SimpleGCDEC(50, 50, device=dev)
        m.record_epoch_times = record
        last_model[0] = m
        torch.cuda.synchronize()
        if record:  # no per-launch events next to the per-iteration ones
            m.fit_with_init(emb, adj, init_y, lr=0.005, epochs=epochs, update_interval=3, opt="admin")
            torch.cuda.synchronize()
            return None, None
        with kernels.KernelTimer() as timer:
            t0 = time.perf_counter()
            m.fit_with_init(emb, adj, init_y, lr=0.005, epochs=epochs, update_interval=3, opt="admin")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        return dt, _kernel_totals(timer)
    fit(e1)
    t_a, k_a = fit(e1)
    t_b, k_b = fit(e2)
    units = e2 - e1
    # device time of iterations e1 .. e2 - 1 of one more fit, by events at the iteration boundaries (a multiple of the update interval, so the
    # target-distribution iterations are in it in proportion).  The wall-clock difference of two fits is kept beside it: with ~15 ms of
    # iterations against ~60 ms of fixed cost per fit it gave 0.17 - 0.58 ms for the same kernels.
    fit(e2, record=True)
    ms = sum(last_model[0].epoch_ms[e1:e2]) / units
    ms_wall = (t_b - t_a) / units * 1e3
    ks = _per_unit(k_b, k_a, units)
    dom = max(ks, key=ks.get) if ks else None
    nnz = n * k
    b_min = 2 * (nnz * 8.0 + 4.0 * (n + 1) + 2.0 * n * 50 * 4) + 2 * 2.0 * n * 50 * 4   # two SpMMs by B_min + GEMM fwd + GEMM dW
    roof = {"kernel": dom, "bound": "hbm", "achieved": round(b_min * (1 + 1.0 / 3) / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(b_min * (1 + 1.0 / 3) / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "basis": "whole iteration: SURVEY 8(d)'s 1.85 KB/spot-equivalent for fwd+bwd (B_min) x 4/3 (the target-distribution forward every 3rd epoch), "
                     "DEC head excluded, / iteration time"}
    # CPU: restated SimpleGCDEC + sparse adjacency on a sample
    gs, _ = _spatial_graph(cpu_spots, k, dev)
    adj_s = SpaGCN(l=1.2, device=dev).calc_adj_exp(gs)
    rows = torch.arange(cpu_spots).repeat_interleave(k)
    a_cpu = torch.sparse_coo_tensor(torch.stack([rows, adj_s.col.cpu().long()]), adj_s.val.cpu(), (cpu_spots, cpu_spots))
    torch.manual_seed(0)
    ref = om.SimpleGCDEC(50, 50)
    xs = torch.randn(cpu_spots, 50)
    with torch.no_grad():
        ref.mu = torch.nn.Parameter(torch.randn(10, 50))
        p_t = ref.target_distribution(ref(xs, a_cpu)[1])
    opt = torch.optim.Adam(ref.parameters(), lr=0.005)
    med, it = _cpu_time(lambda: om.spagcn_iteration(ref, opt, xs, a_cpu, p_t), min_seconds=4.0, max_iters=8)
    return {"workload": f"SpaGCN SimpleGCDEC.fit_with_init, one DEC iteration (GraphConvolution 50 -> 50 fwd + Student-t head + KL + bwd + Adam; target "
                        f"distribution every 3rd), {n} spots on a jittered hex grid, spatial kNN k = {k} truncated Gaussian adjacency (the dense N x N of the "
                        f"reference does not exist at this size), fp32; device time of iterations {e1} .. {e2 - 1} of a fit({e2}) / {e2 - e1} (kernels_ms: fit({e2}) - fit({e1}))",
            "ms": round(ms, 4), "value": n / (ms * 1e-3), "unit": "spots/s per iteration", "wall_clock_difference_ms": round(ms_wall, 4), "graph_build_s": round(build_s, 4), "kernels_ms": ks,
            "other_ms": round(ms - sum(ks.values()), 4), "roofline": roof,
            "cpu_baseline": {"value": cpu_spots / med, "unit": "spots/s per iteration", "cores": torch.get_num_threads(), "kind": "port",
                             "sample": f"{cpu_spots} spots of the same generator, oracle.models.SimpleGCDEC + spagcn_iteration on torch-CPU with a sparse "
                                       f"adjacency (the reference's dense adj is N^2), median of {it} ({med * 1e3:.0f} ms each)"}}


def run_all(dev, which=None):
    import gc
    shared = {}

    def scdsc_100k():
        r = c2_scdsc_epoch(dev, 100_000, 1, 6)
        shared["cpu"] = r.get("cpu_baseline")
        return r

    def scdsc_1m():
        cpu = shared.get("cpu")
        if cpu is not None:
            cpu = dict(cpu, sample=cpu["sample"] + " — the rate measured for c2_scdsc_epoch_100k (a per-cell rate of the same port)")
        return c2_scdsc_epoch(dev, 1_000_000, 1, 3, cpu_baseline=cpu)
    table = [("c2_gcn_100k", lambda: c2_gcn_100k(dev)),
             ("c2_scdsc_epoch_100k", scdsc_100k),
             ("c2_scdsc_epoch_1M", scdsc_1m),
             ("c3_scdeepsort_1M_bf16_epoch", lambda: c3_scdeepsort_epoch(dev)),
             ("c4_graphsc_1M_epoch_1gpu", lambda: c4_graphsc_epoch(dev)),
             ("c5_spagcn_500k_iter", lambda: c5_spagcn_iter(dev))]
    out = {}
    for name, fn in table:
        if which and name not in which:
            continue
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 — reported in place; the headline line must survive
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        out[name]["wall_s"] = round(time.perf_counter() - t0, 1)
        gc.collect()
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import json
    names = sys.argv[1:] or None
    print(json.dumps(run_all(torch.device("cuda", 0), names), indent=1))
