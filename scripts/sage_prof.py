"""Development: cycle-stamped timeline of one workgroup of sage_bcm_kernel over a few MFMA steps (library built with -DDH_SB_PROF ->
dance_amd/libdancehip_prof.so): per wave and step the offsets (cycles from the step's first stamp of the workgroup) of
barrier-exit | lag MFMAs issued | clear/scatter done | lead first B fragments | lead MFMAs issued | at the barrier."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import _lib  # noqa: E402
if os.path.exists(os.path.join(os.path.dirname(_lib.LIB_PATH), "libdancehip_prof.so")):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libdancehip_prof.so")  # the -DDH_SB_PROF build
from dance_amd import kernels  # noqa: E402

dev = "cuda"
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_genes, dfeat, per = 2000, 400, 200
g = torch.Generator(device=dev).manual_seed(0)
col = torch.rand(n_cells, n_genes, device=dev, generator=g).topk(per, dim=1).indices.sort(dim=1).values.to(torch.int32)
col = torch.cat((col, (n_genes + torch.arange(n_cells, device=dev, dtype=torch.int32))[:, None]), 1).reshape(-1).contiguous()
rowptr = torch.arange(0, n_cells * (per + 1) + 1, per + 1, dtype=torch.int32, device=dev)
w = torch.rand(col.numel(), device=dev, generator=g) + 0.5
feats = torch.randn(n_genes + n_cells, dfeat, device=dev, generator=g)
cid = torch.cat((torch.arange(n_genes, dtype=torch.int32), -torch.ones(n_cells, dtype=torch.int32))).to(dev)
alpha = torch.rand(n_genes + 2, device=dev, generator=g) + 0.5
lib = _lib.load()
fn = lib.dh_sage_bcm_prof_read
fn.argtypes, fn.restype = [ctypes.c_void_p, ctypes.c_int], ctypes.c_int
STEPS, PROBES = 6, 6
for dt in ("f32", "bf16"):
    h = feats if dt == "f32" else feats.to(torch.bfloat16)
    args = (rowptr, col, w, cid, cid[n_genes:].contiguous(), alpha, h)
    for _ in range(2):
        kernels.sage_aggregate_mfma(*args, 0, n_genes)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * STEPS * PROBES))()
    fn(buf, 0)
    t = [[[buf[(wv * STEPS + s) * PROBES + k] for k in range(PROBES)] for s in range(STEPS)] for wv in range(8)]
    base = min(t[wv][0][0] for wv in range(8))
    print(f"== {dt}: wave (group, column wave) | per step: stamps relative to the first barrier exit; roles: waves 0-3 lead, 4-7 lag; column waves 2,3 move features")
    for wv in range(8):
        rows = []
        for s in range(STEPS):
            rows.append(" ".join(f"{t[wv][s][k] - base:6d}" if t[wv][s][k] else "     -" for k in range(PROBES)))
        print(f"wave {wv} ({wv >> 2},{wv & 3}): " + " | ".join(rows))
