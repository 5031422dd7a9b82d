"""GPU probe: dh_gcn_narrow_forward_f32 alone at BASELINE config 5's size (500k spots, k = 15, 50 -> 50), X rows at stride 64 and 50,
on five graphs that separate what bounds the kernel: the grid-ordered spatial kNN graph, a random graph (every gather an L2 miss), a
15-wide band and a 5 x 3 grid stencil (every gather near), and a graph whose every edge points at row 0 (the issue floor: no memory
system in the way).  Round 5: the round-4 kernel took 0.62 ms on ALL of them — which is how its serialised loads were found."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_configs import _spatial_graph  # noqa: E402
from bench_rows import gpu_ms  # noqa: E402

from dance_amd import kernels  # noqa: E402
from dance_amd.graph import CSRGraph  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    n, k = 500_000, 15
    gen = torch.Generator(device=dev).manual_seed(0)
    gs, _ = _spatial_graph(n, k, dev)
    colk = torch.randint(0, n, (n, k), device=dev, generator=gen).sort(dim=1).values.to(torch.int32).reshape(-1)
    gr = CSRGraph(torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev), colk, torch.full((n * k, ), 1 / 15., device=dev), n, n)
    x64 = torch.zeros(n, 64, device=dev)
    x64[:, :50] = torch.randn(n, 50, device=dev, generator=gen)
    x50 = x64[:, :50].contiguous()
    w = torch.randn(50, 50, device=dev, generator=gen) / 7
    b = torch.zeros(50, device=dev)
    rp = torch.arange(0, n * k + 1, k, dtype=torch.int32, device=dev)
    off = torch.arange(-7, 8, device=dev)
    band = (torch.arange(n, device=dev)[:, None] + off[None, :]).clamp_(0, n - 1).to(torch.int32).reshape(-1)
    gb = CSRGraph(rp, band, torch.full((n * k, ), 1 / 15., device=dev), n, n)          # every gathered row is L1/L2 resident
    band2 = (torch.arange(n, device=dev)[:, None] + 708 * (off[None, :] % 5 - 2) + off[None, :] // 5).clamp_(0, n - 1)
    gb2 = CSRGraph(rp, band2.sort(dim=1).values.to(torch.int32).reshape(-1), torch.full((n * k, ), 1 / 15., device=dev), n, n)  # 5 grid rows x 3
    g0 = CSRGraph(rp, torch.zeros(n * k, dtype=torch.int32, device=dev), torch.full((n * k, ), 1 / 15., device=dev), n, n)  # one row: issue floor
    for gname, g in (("spatial", gs), ("random", gr), ("band15", gb), ("grid5x3", gb2), ("row0", g0)):
        for xname, x in (("ld64", x64[:, :50]), ("ld50", x50)):
            if True:
                ms = gpu_ms(lambda: kernels.gcn_narrow_forward(g.rowptr, g.col, g.val, x, w, b, kernels.ACT_RELU), iters=20, warm=3)
                ms_na = gpu_ms(lambda: kernels.gcn_narrow_forward(g.rowptr, g.col, g.val, x, w, b, kernels.ACT_RELU, want_agg=False), iters=20, warm=3)
                print(f"{gname:8s} {xname}: {ms:.3f} ms  (without the agg store {ms_na:.3f})", flush=True)


if __name__ == "__main__":
    main()
