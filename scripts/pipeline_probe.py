#!/usr/bin/env python
"""Software-pipelined GCN layer (dance_amd/autograd.py PIPELINE): step time of the headline layer for several slice schedules,
GEMM tile configurations and stream priorities, each checked bit for bit against the unpipelined layer.
    python scripts/pipeline_probe.py [cells] > gpurun_out/pipeline_probe.json"""
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
F, H, K = bench.N_GENES, bench.N_HIDDEN, bench.K_NEIGH
rowptr, col, val = bench.synth_rand_graph(n, K, dev, seed=1)
graph = CSRGraph(rowptr, col, val, n, n)
graph.transpose()
x = bench.synth_features(n, F, dev, seed=100)
gen = torch.Generator(device=dev).manual_seed(2)
bound = (6.0 / (F + H))**0.5
w = ((torch.rand((F, H), device=dev, generator=gen) * 2 - 1) * bound).requires_grad_(True)
dy = torch.randn((n, H), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
torch.cuda.synchronize()
out = {}


def run(label, pipeline, tile=kernels.GEMM_TILE_128, side_prio=0, main_high=False, check=None):
    autograd.PIPELINE, autograd.PIPELINE_TILE, autograd.PIPELINE_SIDE_PRIORITY = pipeline, tile, side_prio
    stream = torch.cuda.Stream(device=dev, priority=-1) if main_high else torch.cuda.current_stream(dev)
    res = {}

    def step():
        w.grad = None
        y = autograd.gcn_layer(x, w, graph, None, True)
        y.backward(dy)
        return y

    stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(stream):
        for _ in range(2):
            y = step()
        torch.cuda.synchronize()
        if check is not None:
            res["bit_identical_y"] = bool(torch.equal(y, check[0]))
            res["bit_identical_dw"] = bool(torch.equal(w.grad, check[1]))
            res["dw_max_rel"] = float((w.grad - check[1]).abs().max() / check[1].abs().max())
        with kernels.KernelTimer() as timer:
            t0 = time.perf_counter()
            for _ in range(steps):
                y = step()
            torch.cuda.synchronize()
            res["ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
        res["kernels_ms"] = {k: [v[0] // steps, round(v[1], 3)] for k, v in sorted(timer.summary().items())}
        # without the per-launch events (they serialise nothing, but cost host time per launch)
        t0 = time.perf_counter()
        for _ in range(steps):
            y = step()
        torch.cuda.synchronize()
        res["ms_per_step_untimed"] = (time.perf_counter() - t0) / steps * 1e3
    out[label] = res
    print(label, json.dumps(res), file=sys.stderr, flush=True)
    return y.detach().clone(), w.grad.detach().clone()


ref = run("off (256x256 tiles, serial)", "off")
run("off, 128x128 tiles", "off")  # PIPELINE off ignores the tile: measured below through the slice list instead
for sched in ("128,128,128,128", "256,256", "256,128,128", "384,128", "128,384"):
    run(f"pipeline {sched} tile128", sched, check=ref)
run("pipeline 128x4 tile128 side-high", "128,128,128,128", side_prio=-1, check=ref)
run("pipeline 128x4 tile128 main-high", "128,128,128,128", main_high=True, check=ref)
run("pipeline 256,256 tile256", "256,256", tile=kernels.GEMM_TILE_256, check=ref)
run("pipeline 256,256 tile-auto", "256,256", tile=kernels.GEMM_TILE_AUTO, check=ref)

# the GEMMs alone in both tile configurations, full width and one 128-column slice
s_buf = torch.empty((n, H), device=dev)


def t_ms(fn, it=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e3


wd = w.detach()
for name, tile in (("256", kernels.GEMM_TILE_256), ("128", kernels.GEMM_TILE_128)):
    out[f"gemm NN full tile{name}"] = t_ms(lambda: kernels.gemm(x, wd, out=s_buf, tile=tile))
    out[f"gemm TN full tile{name}"] = t_ms(lambda: kernels.gemm(x, dy, trans_a=True, tile=tile))
out["gemm NN slice128 tile128"] = t_ms(lambda: kernels.gemm(x, wd[:, :128], out=s_buf[:, :128], tile=kernels.GEMM_TILE_128))
out["gemm TN slice128 tile128"] = t_ms(lambda: kernels.gemm(x, dy[:, :128], trans_a=True, tile=kernels.GEMM_TILE_128))
out["gemm NN slice256 tile128"] = t_ms(lambda: kernels.gemm(x, wd[:, :256], out=s_buf[:, :256], tile=kernels.GEMM_TILE_128))
out["gemm TN slice256 tile128"] = t_ms(lambda: kernels.gemm(x, dy[:, :256], trans_a=True, tile=kernels.GEMM_TILE_128))
out["gemm NN slice256 tile256"] = t_ms(lambda: kernels.gemm(x, wd[:, :256], out=s_buf[:, :256], tile=kernels.GEMM_TILE_256))
print(json.dumps(out, indent=1))
