"""GPU: dh_sage_window_splitk at the 1M-cell size — 2000 gene rows x 1e6 cell columns at 10 % density, D = 400 — timed (plan apart)
and checked against the gather kernel.  Writes gpurun_out/$TAG/splitk.json."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import _lib  # noqa: E402
PROF = os.environ.get("PROF") == "1" and os.path.exists(os.path.join(os.path.dirname(_lib.LIB_PATH), "libdancehip_prof.so"))
if PROF:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libdancehip_prof.so")  # the -DDH_SB_PROF build (scripts/build_prof.sh)
if os.environ.get("VARIANT"):  # A/B builds of sage_bcm.hip: dance_amd/libdancehip_<VARIANT>.so
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libdancehip_{os.environ['VARIANT']}.so")
from dance_amd import kernels  # noqa: E402

dev = "cuda"
n_cells, n_genes, width = int(os.environ.get("CELLS", 1_000_000)), 2000, 400
g = torch.Generator(device=dev).manual_seed(0)
counts, cols = [], []
for g0 in range(0, n_genes, 100):
    m = torch.rand(100, n_cells, device=dev, generator=g) < 0.1
    counts.append(m.sum(1))
    cols.append(m.nonzero()[:, 1].to(torch.int32) + n_genes)
counts = torch.cat(counts)
rowptr = torch.zeros(n_genes + 1, dtype=torch.int64, device=dev)
rowptr[1:] = torch.cumsum(counts + 1, 0)
nnz = int(rowptr[-1])
col = torch.empty(nnz, dtype=torch.int32, device=dev)
is_self = torch.zeros(nnz, dtype=torch.bool, device=dev)
is_self[rowptr[:-1]] = True
col[is_self] = torch.arange(n_genes, dtype=torch.int32, device=dev)
col[~is_self] = torch.cat(cols)
del cols, is_self
w = torch.rand(nnz, device=dev, generator=g) + 0.25
rowptr = rowptr.to(torch.int32)
cid = torch.cat((torch.randperm(n_genes, device=dev, generator=g).to(torch.int32), -torch.ones(n_cells, dtype=torch.int32, device=dev)))
dst = cid[:n_genes].contiguous()
alpha = torch.rand(n_genes + 2, device=dev, generator=g) + 0.5
h = torch.randn(n_genes + n_cells, width, device=dev, generator=g)
out = {"cells": n_cells, "genes": n_genes, "width": width, "entries": nnz}


def gpu_ms(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


torch.cuda.synchronize()
t0 = time.perf_counter()
kernels._sage_plan(rowptr, col, w, n_genes, n_cells, "splitk")
torch.cuda.synchronize()
out["plan_ms_first_call"] = (time.perf_counter() - t0) * 1e3
for tag, hh in (("f32", h), ("bf16", h.to(torch.bfloat16))):
    args = (rowptr, col, w, cid, dst, alpha, hh)
    with kernels.KernelTimer() as tm:
        got = kernels.sage_aggregate_splitk(*args, n_genes, n_cells)
        torch.cuda.synchronize()
    out[f"{tag}_kernels"] = {k: round(v[0] * v[1], 3) for k, v in tm.summary().items()}
    if PROF:  # timeline of workgroup 100, steps 1000 .. 1005 (see scripts/sage_prof.py for the columns)
        import ctypes
        fn = _lib.load().dh_sage_bcm_prof_read
        fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
        STEPS, PROBES = 6, 6
        buf = (ctypes.c_ulonglong * (8 * STEPS * PROBES))()
        fn(buf, 0)
        t = [[[buf[(wv * STEPS + s) * PROBES + k] for k in range(PROBES)] for s in range(STEPS)] for wv in range(8)]
        base = min(t[wv][0][0] for wv in range(8))
        for wv in range(8):
            print(f"{tag} wave {wv} ({wv >> 2},{wv & 3}): " + " | ".join(
                " ".join(f"{t[wv][s][k] - base:6d}" if t[wv][s][k] else "     -" for k in range(PROBES)) for s in range(STEPS)))
    ref = (kernels.sage_aggregate_bf16 if tag == "bf16" else kernels.sage_aggregate)(*args)
    err = float((got.float() - ref.float()).abs().max() / ref.float().abs().max())
    out[f"{tag}_ms"] = gpu_ms(lambda: kernels.sage_aggregate_splitk(*args, n_genes, n_cells))
    out[f"{tag}_gather_ms"] = gpu_ms(lambda: (kernels.sage_aggregate_bf16 if tag == "bf16" else kernels.sage_aggregate)(*args), iters=2, warm=1)
    out[f"{tag}_max_err_vs_gather"] = err
    print(tag, out[f"{tag}_ms"], out[f"{tag}_gather_ms"], err, out[f"{tag}_kernels"], flush=True)
tagdir = os.path.join("gpurun_out", os.environ.get("TAG", "splitk"))
os.makedirs(tagdir, exist_ok=True)
json.dump(out, open(os.path.join(tagdir, "splitk.json"), "w"), indent=1)
print(json.dumps(out))
