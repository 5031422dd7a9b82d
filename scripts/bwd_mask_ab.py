"""A/B of the ReLU mask in the backward SpMM at the headline shape (1M x 512, rand-k15): fused into the gather
(dh_spmm_csr_relu_f32, the default) vs dy masked once + the plain SpMM: python scripts/bwd_mask_ab.py [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev, k, d = "cuda", 15, 512
g = torch.Generator(device=dev).manual_seed(0)
col = torch.randint(0, n, (n, k), device=dev, generator=g).sort(dim=1).values.to(torch.int32).reshape(-1)
rowptr = torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev)
val = torch.full((n * k, ), 1.0 / k, device=dev)
rp_t, col_t, val_t, _ = kernels.csr_transpose(rowptr, col, val, n, n)
z = torch.randn(n, d, device=dev, generator=g)
dy = torch.randn(n, d, device=dev, generator=g)
mask = torch.empty(kernels.relu_mask_bytes(n, d), dtype=torch.uint8, device=dev)
kernels.spmm_csr_relu(rowptr, col, val, z, n_cols=n, act=kernels.ACT_RELU, out_mask=mask)
ident = torch.arange(n, dtype=torch.int32, device=dev)
scratch = torch.empty(n, d, device=dev)
out = torch.empty(n, d, device=dev)


def ms(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


res = {
    "fused mask in the gather (default)": ms(lambda: kernels.spmm_csr_relu(rp_t, col_t, val_t, dy, n_cols=n, in_mask=mask)),
    "mask pass only (gather_rows, identity index)": ms(lambda: kernels.gather_rows(dy, ident, relu_mask=mask, out=scratch)),
    "plain SpMM of the masked matrix": ms(lambda: kernels.spmm_csr(rp_t, col_t, val_t, scratch, n_cols=n, out=out)),
}
res["premask total"] = res["mask pass only (gather_rows, identity index)"] + res["plain SpMM of the masked matrix"]
a = kernels.spmm_csr_relu(rp_t, col_t, val_t, dy, n_cols=n, in_mask=mask)
b = kernels.spmm_csr(rp_t, col_t, val_t, kernels.gather_rows(dy, ident, relu_mask=mask), n_cols=n)
res["bit_identical"] = bool(torch.equal(a, b))
print(json.dumps(res, indent=1))
