#!/usr/bin/env python
"""Why does the 128 x 128 GEMM lose 13-20 % next to the resident aggregation (profiles/r05a_overlap_probe.json)?
One GEMM stream + one resident-aggregation stream, the aggregation varied:
  * X zero-filled (matrix-core power drops: if the loss is DVFS it shrinks)        -> H1 power / clock
  * gathered rows confined to 2048 rows of Z (L2-resident: no HBM traffic)          -> H2 memory-side latency vs CU-side contention
  * 64 / 128 / 256 resident workgroups                                              -> the exchange rate GEMM ms lost : aggregation done
sclk / power are sampled with rocm-smi while each pair runs in a ~1.5 s loop.
    python scripts/overlap_diag.py [label] > gpurun_out/overlap_diag_<label>.json"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dance_amd import kernels  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "default"
quick = len(sys.argv) > 2 and sys.argv[2] == "quick"
n = 1_000_000
dev = torch.device("cuda", 0)
F, H, K = bench.N_GENES, bench.N_HIDDEN, bench.K_NEIGH
rowptr, col, val = bench.synth_rand_graph(n, K, dev, seed=1)
col_l2 = (col % 2048).contiguous()
x = bench.synth_features(n, F, dev, seed=100)
gen = torch.Generator(device=dev).manual_seed(2)
w = (torch.rand((F, H), device=dev, generator=gen) * 2 - 1) * (6.0 / (F + H))**0.5
s_buf = torch.empty((n, H), device=dev)
s_buf2 = torch.empty((n, H), device=dev)
y_buf = torch.empty((n, H), device=dev)
mask = torch.empty(kernels.relu_mask_bytes(n, H), dtype=torch.uint8, device=dev)
kernels.gemm(x, w, out=s_buf)
main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)
out = {"label": label}


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower() or "fclk" in k.lower()}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def both(tag, xin, cols, resident, tile=kernels.GEMM_TILE_128, R=12, G=2, loops=1, sample=False):
    def agg():
        kernels.spmm_csr_relu(rowptr, cols, val, s_buf, n_cols=n, act=kernels.ACT_RELU, out_mask=mask, out=y_buf, slices=(0, 1), resident=resident)
    res = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
    th = None
    if sample:
        th = threading.Thread(target=sampler)
        th.start()
    for _ in range(loops):
        torch.cuda.synchronize()
        g0, g1, s0, s1 = ev(), ev(), ev(), ev()
        g0.record(main)
        for _ in range(G):
            kernels.gemm(xin, w, out=s_buf2, tile=tile)
        g1.record(main)
        if resident is not False:
            with torch.cuda.stream(side):
                s0.record(side)
                for _ in range(R):
                    agg()
                s1.record(side)
        torch.cuda.synchronize()
    if th is not None:
        stop.set()
        th.join()
        res["smi"] = samples[1:-1][:6]
    res["gemm_ms_each"] = round(g0.elapsed_time(g1) / G, 3)
    if resident is not False:
        res["agg_ms_each"] = round(s0.elapsed_time(s1) / R, 3)
        res["agg_done_during_gemm"] = round(g0.elapsed_time(g1) / (s0.elapsed_time(s1) / R), 2)
    out[tag] = res
    print(tag, json.dumps(res), file=sys.stderr, flush=True)


x0 = torch.zeros_like(x)
L = 1 if quick else 25
both("gemm alone", x, col, False, loops=L, sample=not quick)
both("gemm alone (X = 0)", x0, col, False, loops=L, sample=not quick)
for wgs in (64, 128, 256):
    both(f"gemm + resident shape0 wgs{wgs}", x, col, (wgs, 0), R=max(4, 12 * wgs // 256), loops=L if wgs == 256 else 1, sample=(wgs == 256 and not quick))
both("gemm (X = 0) + resident shape0 wgs256", x0, col, (256, 0), loops=L, sample=not quick)
both("gemm + resident shape0 wgs256, L2-resident gather", x, col_l2, (256, 0), R=24)
both("gemm + resident shape1 wgs256, L2-resident gather", x, col_l2, (256, 1), R=24)
both("gemm tile256 + resident shape0 wgs256", x, col, (256, 0), tile=kernels.GEMM_TILE_256)
both("gemm tile256 alone", x, col, False, tile=kernels.GEMM_TILE_256)
print(json.dumps(out, indent=1))
