"""The long-and-narrow bf16 products of the C3 dense update (1M x 400 -> 200 forward with bias + ReLU, its dX 1M x 200 -> 400), ms and
fraction of HBM peak (bytes = A + C once): python scripts/gemm_bf16_rows_time.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402


def timed(fn, it=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


bf, out = torch.bfloat16, {}
M = 1_000_000
for name, K, N, tb in (("fwd 1M x 400 -> 200, bias + ReLU, bf16 out", 400, 200, True), ("dX 1M x 200 -> 400 (W row-major [200, 400])", 200, 400, False),
                       ("1M x 512 -> 256", 512, 256, True), ("1M x 256 -> 64", 256, 64, True), ("1M x 56 -> 32", 56, 32, True)):
    A = torch.randn(M, K, device="cuda").to(bf)
    B = (torch.randn(N, K, device="cuda") if tb else torch.randn(K, N, device="cuda")).to(bf)
    bias = torch.randn(N, device="cuda")
    C = torch.empty(M, N, device="cuda", dtype=bf)
    try:
        ms = timed(lambda: kernels.gemm_bf16(A, B, trans_b=tb, bias=bias, act=kernels.ACT_RELU, out=C))
    except Exception as e:  # noqa: BLE001
        out[name] = {"error": str(e)[-160:]}
        continue
    byt = M * (K + N) * 2
    ref = torch.relu(A[:4096].float() @ (B.float().T if tb else B.float()) + bias)
    err = float((C[:4096].float() - ref).abs().max() / ref.abs().max())
    out[name] = {"ms": round(ms, 4), "GBps": round(byt / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / 8000, 3), "TFLOPs": round(2 * M * K * N / ms / 1e9, 1),
                 "max_err_vs_torch_first_4096_rows": err}
print(json.dumps(out, indent=1))
