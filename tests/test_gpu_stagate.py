"""GPU: STAGATE's GATConv (edge softmax + CSR SpMM with a hand-written backward) and the Stagate auto-encoder against
tests/golden/stagate.npz — the reference's OWN GATConv / Stagate (stagate.py:31-330), AST-lifted and run on torch-CPU over
the restated torch_geometric MessagePassing / softmax (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "stagate.npz")
DEV = "cuda"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def test_gatconv_forward_backward_vs_reference(cuda_device, gold):
    from dance_amd.modules.spatial.spatial_domain.stagate import GATConv
    g = gold
    d, c = g["sg_conv_lin"].shape
    conv = GATConv(d, c, heads=1, concat=False, dropout=0, add_self_loops=False, bias=False).to(DEV)
    with torch.no_grad():
        conv.lin_src.copy_(torch.from_numpy(g["sg_conv_lin"]))
        conv.att_src.copy_(torch.from_numpy(g["sg_conv_att_src"]))
        conv.att_dst.copy_(torch.from_numpy(g["sg_conv_att_dst"]))
    x = torch.from_numpy(g["sg_x"]).to(DEV).requires_grad_(True)
    ei = torch.from_numpy(g["sg_edge_index"]).to(DEV)
    y, (ei2, alpha) = conv(x, ei, return_attention_weights=True)
    assert torch.equal(ei2, ei)
    assert rel_err(y.detach().cpu().numpy(), g["sg_conv_out"]) < 1e-5
    assert rel_err(alpha.detach().cpu().numpy(), g["sg_conv_alpha"]) < 1e-5      # per-edge attention, in the caller's edge order
    y.backward(torch.from_numpy(g["sg_conv_dy"]).to(DEV))
    assert rel_err(x.grad.cpu().numpy(), g["sg_conv_dx"]) < 1e-4
    assert rel_err(conv.lin_src.grad.cpu().numpy(), g["sg_conv_dlin"]) < 1e-4
    assert rel_err(conv.att_src.grad.cpu().numpy(), g["sg_conv_datt_src"]) < 1e-4
    assert rel_err(conv.att_dst.grad.cpu().numpy(), g["sg_conv_datt_dst"]) < 1e-4


def test_stagate_forward_and_pretrain_vs_reference(cuda_device, gold):
    from dance_amd.modules.spatial.spatial_domain.stagate import Stagate
    g = gold
    d = g["sg_x"].shape[1]
    m = Stagate([d, 12, 6], device="cuda")
    sd = {k.split("::", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sg_sd0::")}
    assert sorted(sd) == sorted(m.state_dict())
    m.load_state_dict(sd)
    x, ei = torch.from_numpy(g["sg_x"]).to(DEV), torch.from_numpy(g["sg_edge_index"]).to(DEV)
    with torch.no_grad():
        h2, h4 = m(x, ei)
    assert rel_err(h2.cpu().numpy(), g["sg_h2"]) < 1e-5 and rel_err(h4.cpu().numpy(), g["sg_h4"]) < 1e-5
    # (no reload here: the forward above has re-pointed conv3 / conv4's weights at transposed VIEWS of conv2 / conv1's storage,
    # stagate.py:191-194 — the golden's pre-training started from exactly that state)
    m.pretrain(g["sg_x"], g["sg_edge_index"], lr=1e-2, weight_decay=1e-4, epochs=5, gradient_clipping=5)
    assert rel_err(m.rep, g["sg_rep"]) < 5e-3                      # 5 Adam steps at lr 1e-2 apart from fp32-CPU
    m.fit((g["sg_x"], g["sg_edge_index"]), epochs=2, num_cluster=3, random_state=0)
    assert m.predict().shape == (g["sg_x"].shape[0], ) and set(m.predict()) <= {0, 1, 2}


def test_edge_softmax_kernels_vs_float64(cuda_device):
    """dh_edge_softmax_f32 / _backward on rows of very different length (0, 1, 300 edges), sigmoid and leaky-relu logits."""
    from dance_amd import kernels
    rng = np.random.default_rng(0)
    deg = np.array([0, 1, 300, 7, 64, 65] + list(rng.integers(0, 20, 200)))
    n = deg.size
    rowptr = np.concatenate(([0], np.cumsum(deg))).astype(np.int32)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int32)
    a_src, a_dst = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    datt = rng.standard_normal(col.size).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    rows = np.repeat(np.arange(n), deg)
    for act, f in ((kernels.ATT_SIGMOID, torch.sigmoid), (kernels.ATT_LEAKY_RELU, lambda v: torch.nn.functional.leaky_relu(v, 0.2))):
        att = kernels.edge_softmax(t(rowptr), t(col), t(a_src), t(a_dst), act=act, negative_slope=0.2)
        s64, d64 = torch.from_numpy(a_src).double().requires_grad_(True), torch.from_numpy(a_dst).double().requires_grad_(True)
        e = f(s64[torch.from_numpy(col).long()] + d64[torch.from_numpy(rows)])
        ex = (e - torch.zeros(n, dtype=torch.float64).scatter_reduce(0, torch.from_numpy(rows), e, reduce="amax", include_self=False)[torch.from_numpy(rows)]).exp()
        ref = ex / (torch.zeros(n, dtype=torch.float64).index_add_(0, torch.from_numpy(rows), ex)[torch.from_numpy(rows)] + 1e-16)
        assert rel_err(att.cpu().numpy(), ref.detach().numpy()) < 1e-6
        (ref * torch.from_numpy(datt).double()).sum().backward()
        dt, d_dst = kernels.edge_softmax_backward(t(rowptr), t(col), t(a_src), t(a_dst), att, t(datt), act=act, negative_slope=0.2)
        d_src = torch.zeros(n, device=DEV).index_add_(0, t(col).long(), dt)
        assert rel_err(d_dst.cpu().numpy(), d64.grad.numpy()) < 1e-5 and rel_err(d_src.cpu().numpy(), s64.grad.numpy()) < 1e-5
