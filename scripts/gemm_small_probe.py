"""GPU probe: the products of a graph-sc step at the reference's batch size (128 cells + 2000 genes: block of 2128 source rows) and of a
scDeepSort step at batch 500, through dh_gemm_f32_small (inside kernels.mini_batch_products()) and through the large-tile kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_rows import gpu_ms  # noqa: E402

from dance_amd import kernels  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(0)
    shapes = [("graph-sc conv X W", 2128, 200, 50, False, False), ("graph-sc Linear forward", 128, 300, 200, False, True),
              ("graph-sc Linear dX", 128, 200, 300, False, False), ("graph-sc Linear dW", 300, 200, 128, True, False),
              ("scDeepSort lin forward (fp32)", 500, 200, 400, False, True), ("scDeepSort classifier", 500, 16, 200, False, True),
              ("scDeepSort lin dW", 200, 400, 500, True, False), ("square 1000 x 1000 x 512", 1000, 1000, 512, False, False)]
    for name, M, N, K, ta, tb in shapes:
        a = torch.randn((K, M) if ta else (M, K), device=dev, generator=gen)
        b = torch.randn((N, K) if tb else (K, N), device=dev, generator=gen)
        res = {}
        with kernels.mini_batch_products():
            res[True] = gpu_ms(lambda: kernels.gemm(a, b, trans_a=ta, trans_b=tb), iters=200, warm=20) * 1e3
        res[False] = gpu_ms(lambda: kernels.gemm(a, b, trans_a=ta, trans_b=tb), iters=200, warm=20) * 1e3
        print(f"{name:34s} {M:5d} x {N:5d} x {K:4d}: small {res[True]:6.1f} us   large-tile {res[False]:6.1f} us   (back-to-back launches, eager)", flush=True)


if __name__ == "__main__":
    main()
