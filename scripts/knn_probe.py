"""kNN timing probe: python scripts/knn_probe.py  (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dance_amd import kernels
for n, d in [(100000, 50), (400000, 50), (100000, 16), (100000, 3), (50000, 2000), (20000, 50)]:
    x = torch.randn(n, d, device="cuda") + (100.0 if "--offset" in sys.argv else 0.0)
    for name, algo in (("scan", kernels.KNN_SCAN), ("filter", kernels.KNN_FILTER)):
        if algo == kernels.KNN_SCAN and "--filter-only" in sys.argv:
            continue
        kernels.knn(x, 15, algo=algo); torch.cuda.synchronize()
        with kernels.KernelTimer() as tm:
            t = time.perf_counter(); kernels.knn(x, 15, algo=algo); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(n, d, name, f"{dt*1e3:.1f} ms", f"{3*n*n*d/dt/1e12:.1f} Tops-equivalent", flush=True)
