"""GPU: the fused Student-t soft assignment (dh_student_t_forward_f32 / _backward_f32) against the reference's broadcast formulation
(spagcn.py:394-396, :605-607; scdsc.py:466-468) in float64: q, dZ, dMU; every parameterisation the models use; ragged sizes; the shapes
the kernels refuse fall back to the torch formulation."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ref(z, mu, a, eps, pw, scale):
    q = 1.0 / ((1.0 + torch.sum((z.unsqueeze(1) - mu)**2, dim=2) / a) + eps)
    q = q**pw * scale
    return q / torch.sum(q, dim=1, keepdim=True)


@pytest.mark.parametrize("n,c,d", [(1000, 10, 50), (129, 7, 32), (1, 3, 5), (4097, 32, 40), (300, 20, 60), (128, 2, 1), (700, 17, 33)])
@pytest.mark.parametrize("consts", [dict(a=0.2, eps=1e-8, pw=1.2, scale=0.5), dict(a=0.2, eps=1e-6, pw=1.2, scale=0.5), dict(a=1.0, eps=0.0, pw=1.0, scale=1.0),
                                    dict(a=2.5, eps=0.0, pw=1.75, scale=1.0)])
def test_student_t_matches_float64_formula(cuda_device, n, c, d, consts):
    from dance_amd import kernels
    from dance_amd.autograd import student_t_assign
    assert kernels.student_t_supported(c, d)
    g = torch.Generator(device=cuda_device).manual_seed(n + c + d)
    z = torch.randn(n, d, device=cuda_device, generator=g).requires_grad_(True)
    mu = (torch.randn(c, d, device=cuda_device, generator=g) * 0.7).requires_grad_(True)
    gq = torch.randn(n, c, device=cuda_device, generator=g)
    q = student_t_assign(z, mu, **consts)
    q.backward(gq)
    z64, mu64 = z.detach().double().requires_grad_(True), mu.detach().double().requires_grad_(True)
    q64 = _ref(z64, mu64, **consts)
    q64.backward(gq.double())
    assert rel_err(q.detach().cpu().numpy(), q64.detach().cpu().numpy()) < 1e-5
    assert np.allclose(q.detach().sum(1).cpu().numpy(), 1.0, atol=1e-5)
    assert rel_err(z.grad.cpu().numpy(), z64.grad.cpu().numpy()) < 2e-5
    assert rel_err(mu.grad.cpu().numpy(), mu64.grad.cpu().numpy()) < 2e-5


def test_student_t_strided_rows_and_deterministic(cuda_device):
    """Rows with a leading dimension wider than d (SimpleGCDEC's 256-byte aligned embedding rows); two runs give identical bits."""
    from dance_amd import kernels
    g = torch.Generator(device=cuda_device).manual_seed(3)
    buf = torch.randn(5000, 64, device=cuda_device, generator=g)
    z = buf[:, :50]
    mu = torch.randn(12, 50, device=cuda_device, generator=g)
    gq = torch.randn(5000, 12, device=cuda_device, generator=g)
    q = kernels.student_t_forward(z, mu, 0.2, 1e-8, 1.2, 0.5)
    assert torch.equal(q, kernels.student_t_forward(z.contiguous(), mu, 0.2, 1e-8, 1.2, 0.5))
    dz1, dmu1 = kernels.student_t_backward(z, mu, 0.2, 1e-8, 1.2, 0.5, gq)
    dz2, dmu2 = kernels.student_t_backward(z, mu, 0.2, 1e-8, 1.2, 0.5, gq)
    assert torch.equal(dz1, dz2) and torch.equal(dmu1, dmu2)
    _, dmu3 = kernels.student_t_backward(z, mu, 0.2, 1e-8, 1.2, 0.5, gq, want_dz=False)
    assert torch.equal(dmu1, dmu3)


def test_student_t_unsupported_shapes_fall_back(cuda_device):
    from dance_amd import _lib, kernels
    from dance_amd.autograd import student_t_assign
    assert not kernels.student_t_supported(65, 10) and not kernels.student_t_supported(40, 128) and not kernels.student_t_supported(20, 200)
    z = torch.randn(50, 10, device=cuda_device, requires_grad=True)
    mu = torch.randn(65, 10, device=cuda_device, requires_grad=True)
    q = student_t_assign(z, mu, a=1.0, eps=0.0, pw=1.0, scale=1.0)
    q.sum().backward()
    assert q.shape == (50, 65) and z.grad is not None
    with pytest.raises(_lib.DanceHipError):
        kernels.student_t_forward(z.detach(), mu.detach(), 1.0, 0.0, 1.0, 1.0)
