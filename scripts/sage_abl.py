"""Development: dh_sage_window_mfma at 1M cells x 2000 genes, D = 400 — timing only (VARIANT=<name> loads an A/B build
dance_amd/libdancehip_<name>.so; the second argument is a list of values for the retired DANCE_AMD_SM2_ABL switch, pass 0)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import _lib  # noqa: E402
if os.environ.get("VARIANT"):  # A/B builds of sage_bcm.hip: dance_amd/libdancehip_<VARIANT>.so
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libdancehip_{os.environ['VARIANT']}.so")
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
cid_cells = cid[n_genes:].contiguous()
alpha = torch.rand(n_genes + 2, device=dev, generator=g) + 0.5


def timed(fn, it=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / it, 3)


out = {}
for dt in ("f32", "bf16"):
    h = feats if dt == "f32" else feats.to(torch.bfloat16)
    args = (rowptr, col, w, cid, cid_cells, alpha, h)
    for abl in [int(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,4,6,8,14,15".split(","))]:
        os.environ["DANCE_AMD_SM2_ABL"] = str(abl)
        out[f"{dt} abl={abl}"] = timed(lambda: kernels.sage_aggregate_mfma(*args, 0, n_genes))
    os.environ["DANCE_AMD_SM2_ABL"] = "0"
print(json.dumps(out, indent=1))
