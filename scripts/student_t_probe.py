"""GPU probe: dh_student_t_forward_f32 / _backward_f32 alone at the DEC heads' sizes (SpaGCN: 500k spots x 10 clusters x 50; scDSC: 1M cells
x 10 x 32), against the bytes they move (forward N (d + c) 4, backward N (2 d + 2 c) 4)."""
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
    for n, c, d, consts in ((500_000, 10, 50, (0.2, 1e-8, 1.2, 0.5)), (1_000_000, 10, 32, (1.0, 0.0, 1.0, 1.0)), (500_000, 20, 50, (0.2, 1e-8, 1.2, 0.5)),
                            (500_000, 40, 50, (0.2, 1e-8, 1.2, 0.5))):
        z = torch.randn(n, d, device=dev, generator=gen)
        mu = torch.randn(c, d, device=dev, generator=gen) * 0.7
        g = torch.randn(n, c, device=dev, generator=gen)
        f = gpu_ms(lambda: kernels.student_t_forward(z, mu, *consts), iters=20, warm=3)
        b = gpu_ms(lambda: kernels.student_t_backward(z, mu, *consts, g), iters=20, warm=3)
        fb, bb = n * (d + c) * 4.0, n * (2 * d + 2 * c) * 4.0
        print(f"n {n} c {c} d {d}: forward {f:.3f} ms ({fb / f / 1e6:.0f} GB/s)   backward {b:.3f} ms ({bb / b / 1e6:.0f} GB/s)", flush=True)


if __name__ == "__main__":
    main()
