"""SpMM of the headline layer (1M rows, 512 columns) on the rand-k15 and knn-k15 graphs: plain / ReLU-recording forward on A,
plain / mask-applying backward on A^T.  One JSON object; ms per call (HIP events, 10 calls)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dance_amd import kernels  # noqa: E402
from dance_amd.graph import CSRGraph  # noqa: E402

dev = torch.device("cuda:0")
n, h = 1_000_000, 512


def timed(fn, it=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / it, 4)


z = torch.randn(n, h, device=dev)
dy = torch.randn(n, h, device=dev)
out = torch.empty(n, h, device=dev)
res = {}
for name in ("rand", "knn"):
    if name == "rand":
        rp, c, v = bench.synth_rand_graph(n, 15, dev, seed=1)
        g = CSRGraph(rp, c, v, n, n)
    else:
        g, _ = bench.synth_knn_graph(n, 15, dev, seed=7)
    gt = g.transpose()
    mask = torch.zeros(kernels.relu_mask_bytes(n, h), dtype=torch.uint8, device=dev)
    r = {"nnz": int(g.nnz)}
    r["plain_A"] = timed(lambda: kernels.spmm_csr(g.rowptr, g.col, g.val, z, out=out))
    r["relu_fwd_A"] = timed(lambda: kernels.spmm_csr_relu(g.rowptr, g.col, g.val, z, act=kernels.ACT_RELU, out_mask=mask, out=out))
    r["plain_AT"] = timed(lambda: kernels.spmm_csr(gt.rowptr, gt.col, gt.val, dy, out=out))
    r["masked_bwd_AT"] = timed(lambda: kernels.spmm_csr_relu(gt.rowptr, gt.col, gt.val, dy, in_mask=mask, out=out))
    alg = g.nnz * 8.0 + 4.0 * (n + 1) + g.nnz * h * 4.0 + n * h * 4.0
    r["alg_GB"] = round(alg / 1e9, 2)
    r["relu_fwd_TBs"] = round(alg / r["relu_fwd_A"] / 1e9, 2)
    r["masked_bwd_TBs"] = round(alg / r["masked_bwd_AT"] / 1e9, 2)
    res[name] = r
    del g, gt
print(json.dumps(res, indent=1))
