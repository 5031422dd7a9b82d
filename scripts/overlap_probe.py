#!/usr/bin/env python
"""Does the headline layer's aggregation run BESIDE its GEMM when it takes a fixed footprint?  (round 5)

Part A — co-residency microbenchmark, no dependencies between the two kernels:
    the 128 x 128 GEMM (2 workgroups per CU, 352 of 512 registers per SIMD) alone; the resident aggregation alone (both shapes,
    several grid sizes) and the one-shot grid; then both at once on two streams, in both launch orders.  What the GEMM loses and
    what the aggregation gets while they share the CUs is the whole budget of the overlapped layer.
Part B — the pipelined layer (dance_amd/autograd.py) with the resident kernel beside the GEMM panels, checked bit for bit
    against the serial layer.

    python scripts/overlap_probe.py [cells] [steps] > gpurun_out/overlap_probe.json
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dance_amd import autograd, kernels  # noqa: E402
from dance_amd.graph import CSRGraph  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
parts = sys.argv[3] if len(sys.argv) > 3 else "AB"
dev = torch.device("cuda", 0)
F, H, K = bench.N_GENES, bench.N_HIDDEN, bench.K_NEIGH
rowptr, col, val = bench.synth_rand_graph(n, K, dev, seed=1)
graph = CSRGraph(rowptr, col, val, n, n)
gt = graph.transpose()
x = bench.synth_features(n, F, dev, seed=100)
gen = torch.Generator(device=dev).manual_seed(2)
bound = (6.0 / (F + H))**0.5
w = ((torch.rand((F, H), device=dev, generator=gen) * 2 - 1) * bound).requires_grad_(True)
dy = torch.randn((n, H), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
torch.cuda.synchronize()
out = {"cells": n, "steps": steps}


def log(k, v):
    out[k] = v
    print(k, json.dumps(v), file=sys.stderr, flush=True)


def t_ms(fn, it=4):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / it * 1e3, 3)


wd = w.detach()
s_buf = torch.empty((n, H), device=dev)
y_buf = torch.empty((n, H), device=dev)
mask = torch.empty(kernels.relu_mask_bytes(n, H), dtype=torch.uint8, device=dev)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)

if "A" in parts:
    kernels.gemm(x, wd, out=s_buf)
    ref_y = kernels.spmm_csr_relu(rowptr, col, val, s_buf, n_cols=n, act=kernels.ACT_RELU, out_mask=mask)
    ref_mask = mask.clone()
    a = {}
    a["gemm NN full tile128 alone"] = t_ms(lambda: kernels.gemm(x, wd, out=s_buf, tile=kernels.GEMM_TILE_128))
    a["gemm NN full tile256 alone"] = t_ms(lambda: kernels.gemm(x, wd, out=s_buf, tile=kernels.GEMM_TILE_256))
    a["gemm TN full tile128 alone"] = t_ms(lambda: kernels.gemm(x, dy, trans_a=True, tile=kernels.GEMM_TILE_128))
    a["gemm NN slice128 tile128 alone"] = t_ms(lambda: kernels.gemm(x, wd[:, :128], out=s_buf[:, :128], tile=kernels.GEMM_TILE_128))
    a["gemm NN slice256 tile128 alone"] = t_ms(lambda: kernels.gemm(x, wd[:, :256], out=s_buf[:, :256], tile=kernels.GEMM_TILE_128))
    kernels.gemm(x, wd, out=s_buf)

    def agg(resident, sl=(0, 1), in_mask=None):
        return kernels.spmm_csr_relu(rowptr, col, val, s_buf, n_cols=n, act=kernels.ACT_RELU if in_mask is None else kernels.ACT_NONE,
                                     out_mask=mask if in_mask is None else None, in_mask=in_mask, out=y_buf, slices=sl, resident=resident)

    a["agg slice one-shot alone"] = t_ms(lambda: agg(None))
    a["agg 4 slices one-shot alone"] = t_ms(lambda: agg(None, (0, 4)))
    a["agg slice one-shot masked-in alone"] = t_ms(lambda: agg(None, in_mask=ref_mask))
    for shape in (0, 1):
        for wgs in (256, 512, 1024):
            a[f"agg slice resident shape{shape} wgs{wgs} alone"] = t_ms(lambda: agg((wgs, shape)))
        a[f"agg slice resident shape{shape} wgs256 masked-in alone"] = t_ms(lambda: agg((256, shape), in_mask=ref_mask))
    # bit identity of the resident forms (all four slices, both shapes, sign mask included)
    for shape in (0, 1):
        y_buf.zero_()
        mask.zero_()
        agg((256, shape), (0, 4))
        torch.cuda.synchronize()
        a[f"resident shape{shape} == one-shot (Y, mask)"] = [bool(torch.equal(y_buf, ref_y)), bool(torch.equal(mask, ref_mask))]
    g_in = kernels.spmm_csr_relu(gt.rowptr, gt.col, gt.val, dy, n_cols=n, in_mask=ref_mask)
    for shape in (0, 1):
        g2 = torch.zeros_like(g_in)
        kernels.spmm_csr_relu(gt.rowptr, gt.col, gt.val, dy, n_cols=n, in_mask=ref_mask, out=g2, slices=(0, 4), resident=(256, shape))
        torch.cuda.synchronize()
        a[f"resident shape{shape} masked-in == one-shot"] = bool(torch.equal(g2, g_in))
    del g_in, g2
    log("A alone", a)

    # both at once.  R aggregation launches on the side stream, G GEMMs on the main one; every launch bracketed by events.
    def both(label, resident, first, tile=kernels.GEMM_TILE_128, R=16, G=2, in_mask=None):
        res = {}
        torch.cuda.synchronize()
        ev = lambda: torch.cuda.Event(enable_timing=True)
        g0, g1, s0, s1 = ev(), ev(), ev(), ev()
        t0 = time.perf_counter()

        def run_gemm():
            g0.record(main)
            for _ in range(G):
                kernels.gemm(x, wd, out=s_buf2, tile=tile)
            g1.record(main)

        def run_agg():
            with torch.cuda.stream(side):
                s0.record(side)
                for _ in range(R):
                    agg(resident, in_mask=in_mask)
                s1.record(side)

        if first == "gemm":
            run_gemm(); run_agg()
        else:
            run_agg(); run_gemm()
        torch.cuda.synchronize()
        res["wall_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        res["gemm_ms_each"] = round(g0.elapsed_time(g1) / G, 3)
        res["agg_ms_each"] = round(s0.elapsed_time(s1) / R, 3)
        res["agg_span_ms"] = round(s0.elapsed_time(s1), 3)
        res["gemm_span_ms"] = round(g0.elapsed_time(g1), 3)
        log("A both: " + label, res)

    s_buf2 = torch.empty((n, H), device=dev)
    for first in ("gemm", "agg"):
        both(f"one-shot agg, tile128, {first} first", None, first, R=24)
        both(f"one-shot agg, tile256, {first} first", None, first, tile=kernels.GEMM_TILE_256, R=24)
        for shape in (0, 1):
            both(f"resident shape{shape} wgs256, tile128, {first} first", (256, shape), first, R=12)
        both(f"resident shape0 wgs256 masked-in, tile128, {first} first", (256, 0), first, R=12, in_mask=ref_mask)
    both("resident shape0 wgs512, tile128, gemm first", (512, 0), "gemm", R=12)
    both("resident shape0 wgs256, tile256, gemm first", (256, 0), "gemm", tile=kernels.GEMM_TILE_256, R=12)
    del s_buf2, ref_y

if "B" in parts:
    del s_buf, y_buf

    def run(label, pipeline, resident=None, tile=kernels.GEMM_TILE_128, check=None, bwd_mask=None):
        autograd.PIPELINE, autograd.PIPELINE_TILE, autograd.PIPELINE_RESIDENT = pipeline, tile, resident
        if bwd_mask is not None:
            autograd.BWD_MASK_MODE = bwd_mask
        res = {}

        def step():
            w.grad = None
            y = autograd.gcn_layer(x, w, graph, None, True)
            y.backward(dy)
            return y

        for _ in range(2):
            y = step()
        torch.cuda.synchronize()
        if check is not None:
            res["bit_identical_y"] = bool(torch.equal(y, check[0]))
            res["bit_identical_dw"] = bool(torch.equal(w.grad, check[1]))
            res["dw_max_rel"] = float((w.grad - check[1]).abs().max() / check[1].abs().max())
        t0 = time.perf_counter()
        for _ in range(steps):
            y = step()
        torch.cuda.synchronize()
        res["ms_per_step"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
        with kernels.KernelTimer() as timer:
            for _ in range(3):
                y = step()
            torch.cuda.synchronize()
        res["kernels_ms"] = {k: [v[0] // 3, round(v[1], 3)] for k, v in sorted(timer.summary().items())}
        log("B " + label, res)
        return y.detach().clone(), w.grad.detach().clone()

    ref = run("serial (off)", "off")
    run("serial, fused bwd mask", "off", bwd_mask="fused", check=ref)
    autograd.BWD_MASK_MODE = "premask"
    for sched in ("128,128,128,128", "256,256", "128,128,256"):
        run(f"pipeline {sched} one-shot", sched, None, check=ref)
        for shape in (0, 1):
            run(f"pipeline {sched} resident shape{shape} wgs256", sched, (256, shape), check=ref)
    run("pipeline 128x4 resident shape0 wgs512", "128,128,128,128", (512, 0), check=ref)

print(json.dumps(out, indent=1))
