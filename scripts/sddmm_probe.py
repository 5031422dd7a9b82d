"""GPU probe: dh_sddmm_csr_f32 / _bf16 at 1M rows, k = 15 (random graph and a 15-wide band), against the bytes it has to move
(per edge: one V row + 4 B column + 4 B result; per row: one U row + 4 B pointer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_rows import gpu_ms  # noqa: E402

from dance_amd import kernels  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n, k = 1_000_000, 15
    gen = torch.Generator(device=dev).manual_seed(0)
    rp = torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev)
    rand = torch.randint(0, n, (n, k), device=dev, generator=gen).sort(dim=1).values.to(torch.int32).reshape(-1)
    off = torch.arange(-7, 8, device=dev)
    band = (torch.arange(n, device=dev)[:, None] + off[None, :]).clamp_(0, n - 1).to(torch.int32).reshape(-1)
    for width in (32, 64, 256):
        u = torch.randn(n, width, device=dev, generator=gen)
        v = torch.randn(n, width, device=dev, generator=gen)
        for gname, col in (("random", rand), ("band15", band)):
            for dt in (torch.float32, torch.bfloat16):
                uu, vv = u.to(dt), v.to(dt)
                ms = gpu_ms(lambda: kernels.sddmm_csr(rp, col, uu, vv), iters=10, warm=2)
                es = uu.element_size()
                gathered = n * k * (width * es + 8.0) + n * (width * es + 4.0)   # every V row fetched per edge (random graph: no reuse)
                minimal = n * k * 8.0 + 2.0 * n * width * es + 4.0 * n          # every V row fetched once (what locality allows)
                print(f"width {width:4d} {gname:7s} {str(dt)[6:]:9s}: {ms:.3f} ms   gathered {gathered / ms / 1e6:7.0f} GB/s   minimal {minimal / ms / 1e6:7.0f} GB/s",
                      flush=True)


if __name__ == "__main__":
    main()
