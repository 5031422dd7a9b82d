"""GPU: the DEC heads' fused target distribution and KL loss (csrc/dec_loss.hip) against the reference's torch expressions
(dance/modules/spatial/spatial_domain/spagcn.py:398-407, 421-425) in float64 — values and the gradient with respect to q."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c,ld", [(1000, 10, 10), (37, 3, 5), (200_000, 10, 10), (513, 64, 64), (1, 7, 7), (4097, 20, 24)])
def test_dec_target_and_kl_vs_float64(cuda_device, n, c, ld):
    from dance_amd.autograd import dec_kl_loss, dec_target_distribution
    torch.manual_seed(n + c)
    base = torch.rand(n, ld, device=cuda_device) + 1e-3
    q = base[:, :c] / base[:, :c].sum(1, keepdim=True)
    if ld == c:
        q = q.contiguous()
    p = dec_target_distribution(q)
    qd = q.double()
    pr = qd**2 / qd.sum(0)
    pr = pr / pr.sum(1, keepdim=True)
    assert rel_err(p.cpu().numpy(), pr.cpu().numpy()) < 2e-6
    # the KL loss of the next forward's q (another draw), mean over spots, and its gradient
    q2 = (torch.rand(n, c, device=cuda_device) + 1e-3)
    q2 = (q2 / q2.sum(1, keepdim=True)).requires_grad_(True)
    loss = dec_kl_loss(p, q2, eps=1e-6, scale=1.0 / n)
    g, = torch.autograd.grad(loss * 3.0, q2)
    q2d = q2.detach().double().requires_grad_(True)
    ref = torch.mean(torch.sum(pr * torch.log(pr / (q2d + 1e-6)), dim=1))
    gr, = torch.autograd.grad(ref * 3.0, q2d)
    assert abs(float(loss) - float(ref)) < 2e-6 * max(abs(float(ref)), 1e-3)
    assert rel_err(g.cpu().numpy(), gr.cpu().numpy()) < 2e-6
    l2 = dec_kl_loss(p, q2, eps=1e-6, scale=1.0 / n)
    assert torch.equal(l2, loss)  # fixed summation order


@pytest.mark.skipif(False, reason="")
def test_dec_paths_fall_back_off_gpu():
    from dance_amd.autograd import dec_kl_loss, dec_target_distribution
    q = torch.rand(50, 4) + 0.1
    q = q / q.sum(1, keepdim=True)
    p = dec_target_distribution(q)
    pr = q**2 / q.sum(0)
    assert torch.allclose(p, pr / pr.sum(1, keepdim=True))
    assert torch.allclose(dec_kl_loss(p, q, eps=1e-6, scale=1 / 50), torch.mean(torch.sum(p * torch.log(p / (q + 1e-6)), dim=1)))
