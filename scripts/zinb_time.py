"""dh_zinb_nll_forward/backward_f32 at n x 2000: the dense-ish count matrix of scripts/bench_rows.py (Poisson(U(0, 2)): 57 % non-zero) and a
10 % dense one; milliseconds per forward + backward and the error against the float64 formula on a sample."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cpu_ops  # noqa: E402
from dance_amd import _lib  # noqa: E402
if os.environ.get("VARIANT"):  # A/B builds: dance_amd/libdancehip_<VARIANT>.so
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libdancehip_{os.environ['VARIANT']}.so")
from dance_amd import autograd, kernels  # noqa: E402

dev = "cuda"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gz = 2000
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for name, lam_hi, keep in (("poisson(U(0,2)) 57% non-zero", 2.0, 1.0), ("10% non-zero", 2.0, 0.1755)):
    xr = torch.poisson(torch.rand(n, gz, device=dev, generator=g) * lam_hi)
    if keep < 1:
        xr = xr * (torch.rand(n, gz, device=dev, generator=g) < keep)
    mean = (torch.rand(n, gz, device=dev, generator=g) * 4 + 1e-3).requires_grad_(True)
    disp = (torch.rand(n, gz, device=dev, generator=g) * 3 + 1e-3).requires_grad_(True)
    pi = (torch.rand(n, gz, device=dev, generator=g) * 0.98 + 0.01).requires_grad_(True)
    sf = torch.rand(n, device=dev, dtype=torch.float64) + 0.5

    def step():
        mean.grad = disp.grad = pi.grad = None
        autograd.zinb_nll(xr, mean, disp, pi, sf).backward()

    step()
    torch.cuda.synchronize()
    with kernels.KernelTimer() as tm:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            step()
        b.record()
        torch.cuda.synchronize()
    ns = 20_000
    m_, d_, p_ = (t[:ns].detach().double().requires_grad_(True) for t in (mean, disp, pi))
    ref = cpu_ops._zinb_elements(xr[:ns], m_, d_, p_, sf[:ns], 0.0).mean()
    rm, rd, rp = torch.autograd.grad(ref, (m_, d_, p_))
    ms_, ds_, ps_ = (t[:ns].detach().clone().requires_grad_(True) for t in (mean, disp, pi))
    got = autograd.zinb_nll(xr[:ns].contiguous(), ms_, ds_, ps_, sf[:ns].contiguous())
    gm, gd, gp = torch.autograd.grad(got, (ms_, ds_, ps_))
    rel = lambda x, y: float((x.double() - y).abs().max() / y.abs().max())
    out[name] = dict(ms_fwd_bwd=round(a.elapsed_time(b) / 3, 2), kernels_ms={k: round(v[1], 2) for k, v in tm.summary().items()},
                     nonzero_frac=round(float((xr[:ns] > 0).float().mean()), 3), loss_rel_err=abs(float(got) - float(ref)) / abs(float(ref)),
                     grad_rel_err=dict(mean=rel(gm, rm), disp=rel(gd, rd), pi=rel(gp, rp)),
                     hbm_frac=round(n * gz * 44.0 / (a.elapsed_time(b) / 3) / 1e6 / 8000, 3))
    # the heads' one-pass form on RAW outputs (dh_zinb_heads_fused_f32) next to the logits kernels + the three column sums it replaces
    raws = [torch.randn(n, gz, device=dev, generator=g) for _ in range(3)]
    up = torch.tensor([1.0 / (n * gz)], dtype=torch.float64, device=dev)

    def three_pass():
        kernels.zinb_nll_forward(xr, *raws, sf, 0.0, logits=True)
        for d in kernels.zinb_nll_backward(xr, *raws, sf, 0.0, up, logits=True):
            kernels.colsum(d)

    def timed(fn, it=3):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it

    t3 = timed(three_pass)
    scratch = [r.clone() for r in raws]

    def fused():  # in place: from the second call on the operands are gradients — same shapes, same traffic, other values
        kernels.zinb_heads_fused_(xr, *scratch, sf, 0.0, 1.0 / (n * gz))

    for dst, src in zip(scratch, raws):
        dst.copy_(src)
    with kernels.KernelTimer() as tm2:
        fused()
        torch.cuda.synchronize()
    out[name]["logits_three_pass_ms"] = round(t3, 2)
    out[name]["heads_fused_first_call_kernels_ms"] = {k: round(v[1], 2) for k, v in tm2.summary().items()}
    out[name]["heads_fused_hbm_frac"] = round(n * gz * 28.0 / tm2.summary()["zinb_heads_fused_f32"][1] / 1e6 / 8000, 3)
    del xr, mean, disp, pi, raws, scratch
print(json.dumps(out, indent=1))
