import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels
from dance_amd.autograd import gcn_layer
from dance_amd.graph import CSRGraph
dev = "cuda"; n = 1_000_000
g = torch.Generator(device=dev).manual_seed(0)
colk = torch.randint(0, n, (n, 15), device=dev, generator=g).sort(dim=1).values.to(torch.int32).reshape(-1)
graph = CSRGraph(torch.arange(0, n * 15 + 1, 15, dtype=torch.int32, device=dev), colk, torch.full((n * 15,), 1 / 15., device=dev), n, n)
graph.transpose()
x = torch.randn(n, 50, device=dev, generator=g); w = (torch.randn(50, 50, device=dev, generator=g) / 7).requires_grad_(True)
b = torch.zeros(50, device=dev, requires_grad=True); dy = torch.randn(n, 50, device=dev, generator=g)
def step():
    w.grad = b.grad = None
    gcn_layer(x, w, graph, b, False).backward(dy)
for _ in range(3): step()
torch.cuda.synchronize()
with kernels.KernelTimer() as t:
    for _ in range(10): step()
    torch.cuda.synchronize()
for k, (c, ms) in sorted(t.summary().items()): print(f"{k:28s} {c:3d} x {ms:.3f} ms")
