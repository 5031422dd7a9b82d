#!/usr/bin/env python
"""128 x 128 (two workgroups per CU: one's epilogue under the other's main loop) against 256 x 256 tiles by K: which wins for the
short-K products of the models (ZINB heads 1M x 512 -> 2000, autoencoder layers, GCN chain of scDSC)?
    python scripts/gemm_tile_probe.py > gpurun_out/gemm_tile_probe.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
out = {}


def timed(fn, it=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


M = 1_000_000
for K, N, tb in ((512, 2000, True), (2000, 512, False), (2000, 512, True), (512, 256, True), (256, 256, True), (256, 128, True), (128, 32, True), (512, 256, False),
                 (256, 512, True), (1000, 512, False), (400, 200, True)):
    A = torch.randn(M, K, device=dev, generator=g)
    B = torch.randn((N, K) if tb else (K, N), device=dev, generator=g) / K**0.5
    C = torch.empty(M, N, device=dev)
    row = {}
    for name, tile in (("auto", kernels.GEMM_TILE_AUTO), ("t256", kernels.GEMM_TILE_256), ("t128", kernels.GEMM_TILE_128)):
        try:
            ms = timed(lambda: kernels.gemm(A, B, trans_b=tb, out=C, tile=tile))
            row[name] = {"ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}
        except Exception as e:  # noqa: BLE001
            row[name] = {"error": str(e)[:80]}
    out[f"M={M} K={K} N={N} {'NT' if tb else 'NN'}"] = row
    print(f"K={K} N={N} {'NT' if tb else 'NN'}", json.dumps(row), file=sys.stderr, flush=True)
    del A, B, C
print(json.dumps(out, indent=1))
