import sys, torch
sys.path.insert(0, ".")
from dance_amd import kernels
dev = torch.device("cuda:0")
def t(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for M in (8192, 32768, 131072, 524288, 1000000):
    X = torch.randn(M, 2000, device=dev); W = torch.randn(2000, 512, device=dev); D = torch.randn(M, 512, device=dev)
    f = 2 * M * 2000 * 512
    nn, rb = t(lambda: kernels.gemm(X, W)), t(lambda: torch.mm(X, W))
    tn, rbt = t(lambda: kernels.gemm(X, D, trans_a=True)), t(lambda: torch.mm(X.t(), D))
    print(f"M={M:8d}  NN {f/nn/1e9:6.1f} TF (rocBLAS {f/rb/1e9:6.1f})   TN {f/tn/1e9:6.1f} TF (rocBLAS {f/rbt/1e9:6.1f})")
