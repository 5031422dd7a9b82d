"""dh_gemm_f32x3 at the headline shapes (S = X W: 1M x 2000 x 512 NN; dW = X^T dS: 2000 x 512 x 1M TN) and the NT form: ms, bf16 TFLOP/s of
the six split products and the fraction of the 2.5 PFLOP/s peak; max error against float64 on a sample.  VARIANT=<name> loads an A/B build."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import _lib  # noqa: E402
if os.environ.get("VARIANT"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libdancehip_{os.environ['VARIANT']}.so")
from dance_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
M = 1_000_000
X = torch.randn(M, 2000, device=dev)
W = torch.randn(2000, 512, device=dev) / 45
D = torch.randn(M, 512, device=dev)
Wt = W.t().contiguous()


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


out = {}
for name, fn in (("NN 1M x 2000 x 512", lambda: kernels.gemm(X, W, mode="x3")), ("TN 2000 x 512 x 1M", lambda: kernels.gemm(X, D, trans_a=True, mode="x3")),
                 ("NT 1M x 2000 x 512", lambda: kernels.gemm(X, Wt, trans_b=True, mode="x3"))):
    ms = timed(fn)
    out[name] = {"ms": round(ms, 3), "bf16_TFLOPs": round(6 * 2.0 * M * 2000 * 512 / ms / 1e9, 1), "frac_bf16_peak": round(6 * 2.0 * M * 2000 * 512 / ms / 1e9 / 2500, 3)}
y = kernels.gemm(X[:4096], W, mode="x3")
ref = X[:4096].double() @ W.double()
out["max_rel_err_vs_float64 (4096 rows)"] = float((y.double() - ref).abs().max() / ref.abs().max())
print(json.dumps(out))
