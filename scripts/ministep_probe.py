#!/usr/bin/env python
"""The persistent mini-batch steps at the reference's batch sizes: ms per step of dh_graphsc_steps (batch 128) and dh_scdeepsort_steps
(batch 500, with and without the discarded aggregation, fp32 / bf16 feature storage) on the synthetic cell-gene graph (2000 genes, 10 %
density), HIP events around one C call of `steps` steps; next to it the whole epoch through GraphSC.fit / ScDeepSort.fit.
    python scripts/ministep_probe.py [n_cells=100000] [steps=500] [fit=1]"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import _cellgene_graph  # noqa: E402

n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
with_fit = (sys.argv[3] if len(sys.argv) > 3 else "1") != "0"
batches = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (128, 256, 512)  # graph-sc batch sizes
dev = torch.device("cuda")
out = {"cells": n_cells, "genes": 2000, "edges_per_cell": 200, "steps_per_call": steps}


def time_call(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        fn()
        b.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        if best is None or ms < best[0]:
            best = (ms, t_host * 1e3)
    return best


from dance_amd.ministep import GraphSCStepper, ScDeepSortStepper  # noqa: E402
from dance_amd.modules.single_modality.cell_type_annotation.scdeepsort import GNN, ScDeepSort  # noqa: E402
from dance_amd.modules.single_modality.clustering.graphsc import GraphSC  # noqa: E402

# ---- graph-sc, batch 128 ----------------------------------------------------------------------------------------------------------
cg = _cellgene_graph(n_cells, 2000, 200, 50, dev)
for b in batches:
    n = min(steps, n_cells // b)
    gs = GraphSC(in_feats=50, n_clusters=10, device="cuda")
    gs.model.train()
    opt = torch.optim.Adam(gs.model.parameters(), lr=1e-5, fused=True)
    st = GraphSCStepper(gs.model, cg, b, opt)
    seeds = (2000 + torch.randperm(n_cells, device=dev))[:n * b].contiguous()
    z, loss = torch.empty((n * b, 300), device=dev), torch.empty(n, device=dev)
    ms, host = time_call(lambda: st.run(seeds, n, z, loss))
    st.check_flags("probe")
    out[f"graphsc_b{b}"] = {"ms_per_step": ms / n, "host_ms_per_step": host / n, "steps": n, "loss": float(loss[-1])}
    print(b, out[f"graphsc_b{b}"], file=sys.stderr, flush=True)
if with_fit:
    gs = GraphSC(in_feats=50, n_clusters=10, device="cuda")
    gs.fit(cg, epochs=1, batch_size=128)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gs.fit(cg, epochs=2, batch_size=128)
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    t0 = time.perf_counter()
    gs.fit(cg, epochs=1, batch_size=128)
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    nb = -(-n_cells // 128)
    out["GraphSC.fit_b128"] = {"mode": gs.step_mode, "epoch_s": t2 - t1, "ms_per_step": (t2 - t1) / nb * 1e3, "fit1_s": t1}
    print(out["GraphSC.fit_b128"], file=sys.stderr, flush=True)
del cg

# ---- scDeepSort, batch 500 ----------------------------------------------------------------------------------------------------------
cg = _cellgene_graph(n_cells, 2000, 200, 400, dev)
labels = torch.randint(0, 16, (n_cells, ), generator=torch.Generator().manual_seed(0))
cg.ndata["label"] = torch.cat((-torch.ones(2000, dtype=torch.long), labels)).to(dev)
for tag, bf16, neigh in (("fp32", False, True), ("fp32_no_neigh", False, False), ("bf16", True, True))[:3 if len(batches) > 1 else 1]:
    g = cg.with_ndata(features=cg.ndata["features"].to(torch.bfloat16)) if bf16 else cg
    b = 500
    n = min(steps, n_cells // b)
    model = GNN(400, 16, 200, 1, 2000, activation=torch.nn.ReLU()).to(dev)
    model.train()
    model.layers[0].compute_neigh = neigh
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
    st = ScDeepSortStepper(model, g, b, opt)
    seeds = (2000 + torch.randperm(n_cells, device=dev))[:n * b].contiguous()
    loss = torch.empty(n, device=dev)
    ms, host = time_call(lambda: st.run(seeds, n, loss))
    st.check_flags("probe")
    out[f"scdeepsort_b500_{tag}"] = {"ms_per_step": ms / n, "host_ms_per_step": host / n, "steps": n, "loss_per_cell": float(loss[-1]) / b}
    print(tag, out[f"scdeepsort_b500_{tag}"], file=sys.stderr, flush=True)
if with_fit:
    with tempfile.TemporaryDirectory() as td:
        m = ScDeepSort(400, 200, 1, "mouse", "Brain", batch_size=500, device="cuda", save_root=td, verbose=False)
        m.fit(cg, labels, epochs=1, lr=1e-3, val_ratio=0.2)
        torch.cuda.synchronize()
        ts = {}
        for e in (1, 4):
            t0 = time.perf_counter()
            m.fit(cg, labels, epochs=e, lr=1e-3, val_ratio=0.2)
            torch.cuda.synchronize()
            ts[e] = time.perf_counter() - t0
        # where an epoch of the fit goes (the calls of scdeepsort.py:174-183, one by one)
        graph = m._typed(cg)
        train_idx = (2000 + torch.randperm(n_cells, device=dev))[:int(n_cells * 0.8)]
        parts = {}
        for name, fn in (("cal_loss", lambda: m.cal_loss(graph, train_idx)), ("full_graph_logits", lambda: m._full_graph_logits(graph)),
                         ("evaluate", lambda: m.evaluate(graph, train_idx)), ("save_model", m.save_model)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            parts[name] = (time.perf_counter() - t0) / 3 * 1e3
        out["ScDeepSort.epoch_parts_ms"] = parts
        print(parts, file=sys.stderr, flush=True)
    nb = int(n_cells * 0.8) // 500
    out["ScDeepSort.fit_b500"] = {"mini": bool(m._use_mini), "epoch_s_incl_eval": (ts[4] - ts[1]) / 3, "ms_per_train_batch_incl_eval": (ts[4] - ts[1]) / 3 / nb * 1e3}
    print(out["ScDeepSort.fit_b500"], file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
