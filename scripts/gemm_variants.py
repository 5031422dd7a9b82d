"""A/B timing of GEMM build variants (build/variants/libdh_*.so) against the shipped library; scratch tool."""
import ctypes, glob, os, sys, torch
from ctypes import c_int, c_int64, c_size_t, c_void_p
dev = torch.device("cuda:0")
M, K, N = 1_000_000, 2000, 512
X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) / 45; D = torch.randn(M, N, device=dev)
ref_nn = torch.mm(X[:4096], W); ref_tn = torch.mm(X.t(), D)
libs = {"shipped": "dance_amd/libdancehip.so"}
libs.update({os.path.basename(p)[6:-3]: p for p in sorted(glob.glob("build/variants/libdh_*.so"))})
P, i64 = c_void_p, c_int64
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
rounds = {k: [] for k in libs}
handles = {}
for name, path in libs.items():
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.dh_gemm_f32.restype = c_int
    lib.dh_gemm_f32.argtypes = [i64, i64, i64, c_int, c_int, P, i64, P, i64, P, i64, c_int, P, c_size_t, P]
    lib.dh_gemm_f32_workspace_bytes.restype = c_size_t
    lib.dh_gemm_f32_workspace_bytes.argtypes = [i64, i64, i64, c_int, c_int]
    handles[name] = lib
C1 = torch.empty(M, N, device=dev); C2 = torch.empty(K, N, device=dev)
st = torch.cuda.current_stream().cuda_stream
def nn(lib): assert lib.dh_gemm_f32(M, N, K, 0, 0, X.data_ptr(), K, W.data_ptr(), N, C1.data_ptr(), N, 0, None, 0, st) == 0
def tn(lib):
    wsb = lib.dh_gemm_f32_workspace_bytes(K, N, M, 1, 0)
    ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
    assert lib.dh_gemm_f32(K, N, M, 1, 0, X.data_ptr(), K, D.data_ptr(), N, C2.data_ptr(), N, 0, ws.data_ptr(), wsb, st) == 0
f = 2.0 * M * K * N
for name, lib in handles.items():
    nn(lib); tn(lib); torch.cuda.synchronize()
    e1 = float((C1[:4096] - ref_nn).abs().max()); e2 = float((C2 - ref_tn).abs().max() / ref_tn.abs().max())
    print(name, "err", e1, e2)
for r in range(3):  # interleaved rounds
    for name, lib in handles.items():
        rounds[name].append((timeit(lambda: nn(lib)), timeit(lambda: tn(lib))))
for name, v in rounds.items():
    a = min(x[0] for x in v); b = min(x[1] for x in v)
    print(f"{name:12s} NN {a:6.2f} ms {f/a/1e9:6.1f} TF | TN {b:6.2f} ms {f/b/1e9:6.1f} TF")
