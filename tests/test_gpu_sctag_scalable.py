"""GPU: (f)2 — scTAG's adjacency-decoder loss over all N^2 pairs without an N x N matrix (dh_gram_pairwise_f32 DH_GRAM_SIGMOID_SQ +
dh_sddmm_csr_f32) and the fused ZINB NLL (dh_zinb_nll_forward_f32 / _backward_f32), each against the reference's dense / unfused
formula evaluated in float64 on the device (sctag.py:254,470-471; dance/utils/loss.py:780-829)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

import cpu_ops
from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_gram_pairwise_sigmoid_sq_vs_float64(cuda_device):
    from dance_amd import kernels
    torch.manual_seed(0)
    for n, d in ((257, 8), (1000, 32), (3000, 70), (129, 320)):
        z = torch.randn(n, d, device=cuda_device) * (1.5 / d**0.5)
        rowloss, o = kernels.gram_pairwise(z, kernels.GRAM_SIGMOID_SQ)
        s = torch.sigmoid(z.double() @ z.double().t())
        assert rel_err(rowloss.cpu().numpy(), (s * s).sum(1).cpu().numpy()) < 5e-6
        assert rel_err(o.cpu().numpy(), ((2 * s * s * (1 - s)) @ z.double()).cpu().numpy()) < 2e-5
        r2, o2 = kernels.gram_pairwise(z, kernels.GRAM_SIGMOID_SQ)
        assert torch.equal(r2, rowloss) and torch.equal(o2, o)
        # mode 0 is still the softplus / sigmoid pair
        r0, o0 = kernels.gram_pairwise(z, kernels.GRAM_SOFTPLUS)
        r1, o1 = kernels.gram_sigmoid(z)
        assert torch.equal(r0, r1) and torch.equal(o0, o1)


@pytest.mark.parametrize("n,d,symmetric", [(500, 8, False), (4000, 32, True), (12000, 32, True)])
def test_adj_reconstruction_mse_vs_dense(cuda_device, n, d, symmetric):
    from dance_amd import autograd
    from dance_amd.graph import CSRGraph
    rng = np.random.default_rng(n)
    a = sp.random(n, n, density=12.0 / n, random_state=3, format="csr", dtype=np.float32)
    a.data[:] = rng.uniform(0.2, 1.0, a.nnz)
    if symmetric:
        a = ((a + a.T) * 0.5).tocsr()
    a.sort_indices()
    g = CSRGraph.from_scipy(a, cuda_device, symmetric=symmetric)
    z = (torch.randn(n, d, device=cuda_device) * (1.2 / d**0.5)).requires_grad_(True)
    loss = autograd.adj_reconstruction_mse(z, g)
    gz, = torch.autograd.grad(loss * 3.0, z)
    zd = z.detach().double().requires_grad_(True)
    dense = torch.from_numpy(a.toarray()).to(cuda_device).double()
    ref = F.mse_loss(torch.sigmoid(zd @ zd.t()), dense)
    gref, = torch.autograd.grad(ref * 3.0, zd)
    assert abs(float(loss) - float(ref)) < 2e-6 * abs(float(ref))
    assert rel_err(gz.cpu().numpy(), gref.cpu().numpy()) < 2e-5


def test_zinb_nll_kernels_vs_float64_formula(cuda_device):
    from dance_amd import autograd, kernels
    torch.manual_seed(1)
    for n, g, ridge, with_sf in ((64, 50, 0.0, True), (300, 2000, 0.5, True), (1000, 333, 0.0, False)):
        x = torch.poisson(torch.rand(n, g, device=cuda_device) * 2.5)
        x[:, :5] *= 40  # large counts (> 256): the lgamma / digamma fallback at big arguments
        x[:, 5:9] *= 0.37  # non-integer "counts" (normalised matrices fed by mistake or on purpose): the same fallback at small ones
        mean = (torch.rand(n, g, device=cuda_device) * 5 + 1e-5).requires_grad_(True)
        disp = (torch.rand(n, g, device=cuda_device) * 4 + 1e-4).requires_grad_(True)
        disp.data[:, 7] = 1e4     # the clamp bounds of DispAct
        disp.data[:, 8] = 1e-4
        pi = (torch.rand(n, g, device=cuda_device) * 0.998 + 0.001).requires_grad_(True)
        sf = (torch.rand(n, device=cuda_device, dtype=torch.float64) + 0.5) if with_sf else None
        loss = autograd.zinb_nll(x, mean, disp, pi, sf, ridge)
        assert loss.dtype == torch.float64
        gm, gd, gp = torch.autograd.grad(loss * 2.0, (mean, disp, pi))
        m64, d64, p64 = (t.detach().double().requires_grad_(True) for t in (mean, disp, pi))
        ref = cpu_ops._zinb_elements(x, m64, d64, p64, sf, ridge).mean()
        rm, rd, rp = torch.autograd.grad(ref * 2.0, (m64, d64, p64))
        # (round 4: the x = 0 branch evaluates its log1p / exp / log in fp32, everything else and every sum in float64)
        assert abs(float(loss) - float(ref)) < 1e-6 * abs(float(ref)) + 1e-12
        for a, b, nm in ((gm, rm, "mean"), (gd, rd, "disp"), (gp, rp, "pi")):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6, (n, g, nm)
        again = kernels.zinb_nll_forward(x, mean.detach(), disp.detach(), pi.detach(), sf, ridge)
        assert torch.equal(again, kernels.zinb_nll_forward(x, mean.detach(), disp.detach(), pi.detach(), sf, ridge))


def test_zinb_nll_from_logits_equals_torch_activations_then_loss(cuda_device):
    """dh_zinb_nll_logits_*: the loss on the heads' RAW outputs == MeanAct / DispAct / sigmoid as torch ops (fp32, scdsc.py:601-618)
    followed by the float64 formula, value and gradients w.r.t. the raw outputs — including logits beyond every clamp bound (zero
    gradient there, as torch.clamp's backward gives) and beyond softplus's threshold of 20."""
    from dance_amd import autograd, kernels
    torch.manual_seed(2)
    for n, g, ridge, with_sf in ((64, 50, 0.0, True), (257, 2000, 0.5, True), (1000, 333, 0.0, False)):
        x = torch.poisson(torch.rand(n, g, device=cuda_device) * 2.5) * (torch.rand(n, g, device=cuda_device) < 0.3)
        x[:, :5] *= 40
        am = (torch.randn(n, g, device=cuda_device) * 1.5).requires_grad_(True)
        ad = (torch.randn(n, g, device=cuda_device) * 2.0).requires_grad_(True)
        ap = (torch.randn(n, g, device=cuda_device) * 2.0).requires_grad_(True)
        am.data[:, 9], am.data[:, 10] = -13.0, 14.5      # exp below 1e-5 / above 1e6: clamped, zero gradient
        ad.data[:, 11], ad.data[:, 12], ad.data[:, 13] = -12.0, 25.0, 1.2e4   # softplus below 1e-4; beyond the threshold; above 1e4
        ap.data[:, 14], ap.data[:, 15] = -30.0, 30.0
        sf = (torch.rand(n, device=cuda_device, dtype=torch.float64) + 0.5) if with_sf else None
        loss = autograd.zinb_nll_from_logits(x, am, ad, ap, sf, ridge)
        assert loss.dtype == torch.float64
        gm, gd, gp = torch.autograd.grad(loss * 2.0, (am, ad, ap))
        rm_, rd_, rp_ = (t.detach().clone().requires_grad_(True) for t in (am, ad, ap))
        ref = cpu_ops._zinb_elements(x, *cpu_ops._head_acts(rm_, rd_, rp_), sf, ridge).mean()
        rm, rd, rp = torch.autograd.grad(ref * 2.0, (rm_, rd_, rp_))
        # tolerance: the kernels' exp is v_exp_f32 (relative error |a| 2^-24 <= 8e-7 inside the clamp range), torch's is a full-precision expf
        assert abs(float(loss) - float(ref)) < 2e-6 * abs(float(ref)), (float(loss), float(ref))
        for a, b, nm in ((gm, rm, "mean"), (gd, rd, "disp"), (gp, rp, "pi")):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 5e-6, (n, g, nm)
        for a, cols in ((gm, (9, 10)), (gd, (11, 13))):
            assert bool((a[:, list(cols)] == 0).all())
        # == the two-step form through the library's own kernels on activated inputs (same element code, different exp): 2e-6
        m_, d_, p_ = cpu_ops._head_acts(am.detach(), ad.detach(), ap.detach())
        two = kernels.zinb_nll_forward(x, m_, d_, p_, sf, ridge).sum()
        one = kernels.zinb_nll_forward(x, am.detach(), ad.detach(), ap.detach(), sf, ridge, logits=True)
        assert abs(float(one.sum()) - float(two)) < 2e-6 * abs(float(two))
        assert torch.equal(one, kernels.zinb_nll_forward(x, am.detach(), ad.detach(), ap.detach(), sf, ridge, logits=True))


def test_sctag_scalable_fit_on_device(cuda_device):
    """ScTAG(adj_dim=32) on 6000 cells with a sparse kNN-like adjacency: forward loss == dense formula, pretrain + fit run."""
    from dance_amd.modules.single_modality.clustering.sctag import ScTAG
    rng = np.random.default_rng(0)
    n, g, c = 6000, 200, 4
    lab = rng.integers(0, c, n)
    centers = rng.gamma(1.0, 2.0, (c, g))
    counts = rng.poisson(centers[lab]).astype(np.float32)
    x = np.log1p(counts / counts.sum(1, keepdims=True) * 1e3).astype(np.float32)
    x = ((x - x.mean(0)) / (x.std(0) + 1e-6)).astype(np.float32)
    nbr = np.stack([rng.choice(np.flatnonzero(lab == lab[i]), 10, replace=False) for i in range(n)])
    a = sp.csr_matrix((np.ones(n * 10, np.float32), (np.repeat(np.arange(n), 10), nbr.ravel())), shape=(n, n))
    a = ((a + a.T) > 0).astype(np.float32).tocsr()
    torch.manual_seed(0)
    m = ScTAG(n_clusters=c, k=2, hidden_dim=32, latent_dim=8, dec_dim=[32, 48, 64], dropout=0.0, device="cuda", adj_dim=32)
    m.init_model(a, x)
    z0, z, q, mean, disp, pi = m.forward(m.g_n, torch.from_numpy(x).to(cuda_device))
    loss = m.adj_loss(z0, None)
    zd = z0.detach().double()
    ref = torch.mean(F.mse_loss(torch.sigmoid(zd @ zd.t()), torch.from_numpy(a.toarray()).to(cuda_device).double()))
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    np.random.seed(0)
    m = ScTAG(n_clusters=c, k=2, hidden_dim=32, latent_dim=8, dec_dim=[32, 48, 64], dropout=0.1, device="cuda", adj_dim=32)
    m.fit((a, x, counts, counts.sum(1).astype(np.float64)), lab, epochs=5, pretrain_epochs=10, lr=5e-3)
    from sklearn.metrics import adjusted_rand_score
    assert adjusted_rand_score(lab, m.predict()) > 0.5
