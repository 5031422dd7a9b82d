#!/usr/bin/env python
"""knn-k15 (the metric's literal graph) with and without a locality renumbering: SpMM forward / backward time and the whole layer
step at 1M cells for the orders none / rcm / cluster-sorted (the generator's own labels: an upper bound of what any ordering can
give on this synthetic), each with the rows of the slice kernel mapped round-robin or contiguously onto the XCDs (DH_SPMM_XCDMAP,
read once per process).  Also checks that Y un-permutes bit for bit.
    DH_SPMM_XCDMAP=0|1 python scripts/locality_probe.py [cells] [spmm|all]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dance_amd import autograd, kernels  # noqa: E402
from dance_amd.graph import CSRGraph, locality_order  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
what = sys.argv[2] if len(sys.argv) > 2 else "all"
dev = torch.device("cuda", 0)
F, H, K = bench.N_GENES, bench.N_HIDDEN, bench.K_NEIGH
out = {"xcd_map": os.environ.get("DH_SPMM_XCDMAP", "0"), "cells": n}

g = torch.Generator(device=dev).manual_seed(7)
centers = torch.randn((20, 50), device=dev, generator=g) * 4.0
label = torch.randint(0, 20, (n, ), device=dev, generator=g)
emb = centers[label] + torch.randn((n, 50), device=dev, generator=g)
idx, dist_ = kernels.knn(emb, K)
(rowptr, col, val), _ = kernels.umap_connectivities(idx, dist_.contiguous())
graph = CSRGraph(rowptr, col, val, n, n, symmetric=True)
out["nnz"] = graph.nnz
torch.cuda.synchronize()


def t_ms(fn, it=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


orders = {"none": None}
t0 = time.perf_counter()
orders["rcm"] = locality_order(graph).to(dev)
out["rcm_host_s"] = time.perf_counter() - t0
orders["cluster"] = torch.argsort(label, stable=True)
# rcm inside clusters: clusters contiguous, BFS order within each
s = torch.Generator(device=dev).manual_seed(11)
z = torch.randn((n, H), device=dev, generator=s)
dy = torch.randn((n, H), device=dev, generator=s)
mask = torch.empty(kernels.relu_mask_bytes(n, H), dtype=torch.uint8, device=dev)
y_ref = None
for name, perm in orders.items():
    gp = graph if perm is None else graph.permute(perm)
    zp = z if perm is None else z[perm].contiguous()
    dyp = dy if perm is None else dy[perm].contiguous()
    yp = torch.empty_like(zp)
    r = {}
    r["spmm_fwd_ms"] = t_ms(lambda: kernels.spmm_csr_relu(gp.rowptr, gp.col, gp.val, zp, n_cols=n, act=kernels.ACT_RELU, out_mask=mask, out=yp))
    gbuf = torch.empty_like(dyp)
    r["mask_apply_ms"] = t_ms(lambda: kernels.relu_mask_apply(dyp, mask, out=gbuf))
    r["spmm_bwd_plain_ms"] = t_ms(lambda: kernels.spmm_csr(gp.rowptr, gp.col, gp.val, gbuf, n_cols=n, out=yp))
    r["spmm_bwd_fused_ms"] = t_ms(lambda: kernels.spmm_csr_relu(gp.rowptr, gp.col, gp.val, dyp, n_cols=n, in_mask=mask, out=yp))
    kernels.spmm_csr_relu(gp.rowptr, gp.col, gp.val, zp, n_cols=n, act=kernels.ACT_RELU, out_mask=mask, out=yp)
    if perm is None:
        y_ref = yp.clone()
    else:
        back = torch.empty_like(yp)
        back[perm] = yp
        r["y_bit_identical_after_unpermute"] = bool(torch.equal(back, y_ref))
        del back
    out[f"order={name}"] = r
    print(name, json.dumps(r), file=sys.stderr, flush=True)
    del gp, zp, dyp, yp, gbuf
del z, dy, y_ref
torch.cuda.empty_cache()

# rand-k15 with this mapping (no locality to exploit: must not get slower)
rp, rc, rv = bench.synth_rand_graph(n, K, dev, seed=1)
zz = torch.randn((n, H), device=dev)
yy = torch.empty_like(zz)
out["rand_k15_spmm_fwd_ms"] = t_ms(lambda: kernels.spmm_csr_relu(rp, rc, rv, zz, n_cols=n, act=kernels.ACT_RELU, out_mask=mask, out=yy))
del zz, yy, rp, rc, rv

if what == "all":
    x = bench.synth_features(n, F, dev, seed=100)
    gen = torch.Generator(device=dev).manual_seed(2)
    w = ((torch.rand((F, H), device=dev, generator=gen) * 2 - 1) * (6.0 / (F + H))**0.5).requires_grad_(True)
    dy = torch.randn((n, H), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    ref = None
    for name in ("none", "rcm"):
        perm = orders[name]
        gp = graph if perm is None else graph.permute(perm)
        if perm is not None:  # X permuted ONCE, in place chunk by chunk would need a scratch: here a second 8 GB buffer
            x = x[perm].contiguous()
            dy = dy[perm].contiguous()

        def step():
            w.grad = None
            y = autograd.gcn_layer(x, w, gp, None, True)
            y.backward(dy)
            return y

        y = step()
        torch.cuda.synchronize()
        ms = t_ms(step, 10)
        r = {"ms_per_step": ms, "cells_per_s": n / ms * 1e3}
        if perm is None:
            ref = (y.detach().clone(), w.grad.detach().clone())
        else:
            back = torch.empty_like(y)
            back[perm] = y.detach()
            r["y_bit_identical_after_unpermute"] = bool(torch.equal(back, ref[0]))
            r["dw_max_rel_vs_unpermuted"] = float((w.grad - ref[1]).abs().max() / ref[1].abs().max())
        out[f"layer order={name}"] = r
        print("layer", name, json.dumps(r), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
