"""Achievable HBM bandwidth on this box (SURVEY.md §8d asks for it next to the 8 TB/s spec): device copy and fill."""
import time
import torch
n = 1 << 30  # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def t(fn, iters=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
dt = t(lambda: b.copy_(a)); print(f"copy  (read 4 GiB + write 4 GiB): {2 * n * 4 / dt / 1e12:.2f} TB/s")
dt = t(lambda: b.fill_(1.0)); print(f"fill  (write 4 GiB):             {n * 4 / dt / 1e12:.2f} TB/s")
dt = t(lambda: a.sum()); print(f"sum   (read 4 GiB):              {n * 4 / dt / 1e12:.2f} TB/s")
