"""dh_gram_sigmoid_f32 at graph-sc's large-batch shape (B = 8192 rows, d = 300): ms per call and the fraction of the fp32 matrix-core
peak (VARIANT=<name> loads an A/B build dance_amd/libdancehip_<name>.so)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import _lib  # noqa: E402
if os.environ.get("VARIANT"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libdancehip_{os.environ['VARIANT']}.so")
from dance_amd import kernels  # noqa: E402

out = {}
for b, d in ((8192, 300), (8192, 64), (16384, 300)):
    z = torch.randn(b, d, device="cuda") * (1.5 / d**0.5)
    for _ in range(3):
        kernels.gram_sigmoid(z)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        kernels.gram_sigmoid(z)
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / 20
    out[f"B={b} d={d}"] = {"ms": round(ms, 4), "frac_f32_mfma": round(4.0 * b * b * d / ms / 1e9 / 157.3, 3)}
print(json.dumps(out))
