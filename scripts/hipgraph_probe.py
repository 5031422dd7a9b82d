"""Can a whole training step built from libdancehip launches (ctypes, on torch's current stream) + torch autograd + Adam be captured
as ONE hipGraph (torch.cuda.CUDAGraph) and replayed?  Builds a graph-sc shaped step on static buffers (WeightedGraphConv on a block,
Linear, fused decoder loss, backward, Adam(capturable)), compares the replayed losses with eager, times both.
    python scripts/hipgraph_probe.py [batch=128]"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dance_amd import kernels  # noqa: E402
from dance_amd.autograd import HipLinear, gcn_layer  # noqa: E402
from dance_amd.graph import CSRGraph  # noqa: E402
from dance_amd.modules.single_modality.clustering.graphsc import gram_listed_bce  # noqa: E402

dev = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G, per, d = 2000, 200, 50
out = {"batch": B}
gen = torch.Generator(device=dev).manual_seed(0)
n_src = B + G
col = torch.cat((B + torch.rand(B, G, device=dev, generator=gen).topk(per, dim=1).indices.sort(dim=1).values, torch.arange(B, device=dev)[:, None]), 1)
rowptr = torch.arange(0, B * (per + 1) + 1, per + 1, dtype=torch.int32, device=dev)
val = torch.rand(B * (per + 1), device=dev, generator=gen) + 0.5
graph = CSRGraph(rowptr, col.to(torch.int32).reshape(-1).contiguous(), val, B, n_src)
graph.transpose()
x_static = torch.randn(n_src, d, device=dev, generator=gen)
us = torch.arange(B, device=dev, dtype=torch.int32)
vs = us.clone()
rs = torch.full((B, ), (per + 1)**-0.5, device=dev)
cs = torch.ones(n_src, device=dev)


def make():
    torch.manual_seed(1)
    w = torch.nn.Parameter(torch.randn(d, 200, device=dev) * 0.1)
    b = torch.nn.Parameter(torch.zeros(200, device=dev))
    lin = HipLinear(200, 300).to(dev)
    params = [w, b] + list(lin.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, capturable=True)
    return w, b, lin, opt


def step(w, b, lin, opt, x):
    h = gcn_layer(F.dropout(x, 0.1), w, graph, b, True, rowscale=rs, colscale=cs)
    z = lin(h)
    loss = 0.5 * gram_listed_bce(F.dropout(z, 0.1), us, vs, float(B - 1))
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


try:
    # eager reference trajectory (dropout makes it stochastic: compare with p = 0 via a second run below)
    w, b, lin, opt = make()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eager = [step(w, b, lin, opt, x_static).detach() for _ in range(200)]
    torch.cuda.synchronize()
    out["eager_ms_per_step"] = (time.perf_counter() - t0) / 200 * 1e3
    # capture
    w, b, lin, opt = make()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step(w, b, lin, opt, x_static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = step(w, b, lin, opt, x_static)
    torch.cuda.synchronize()
    out["captured"] = True
    losses = []
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
        losses.append(static_loss.detach().clone())
    torch.cuda.synchronize()
    out["replay_ms_per_step"] = (time.perf_counter() - t0) / 200 * 1e3
    out["loss_first_last_replay"] = [float(losses[0]), float(losses[-1])]
    out["loss_first_last_eager"] = [float(eager[0]), float(eager[-1])]
    out["loss_decreases_under_replay"] = bool(float(losses[-1]) < float(losses[0]))
    # new inputs through the static buffer: the replay must see them
    x_static.copy_(torch.randn(n_src, d, device=dev, generator=gen) * 3)
    g.replay()
    torch.cuda.synchronize()
    out["loss_after_new_input"] = float(static_loss)
except Exception as e:  # noqa: BLE001
    out["captured"] = False
    out["error"] = repr(e)[:600]
print(json.dumps(out, indent=1))
