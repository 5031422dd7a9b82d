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


def _heads_case(n, g, h, dev, with_sf=True):
    x = torch.poisson(torch.rand(n, g, device=dev) * 2.5) * (torch.rand(n, g, device=dev) < 0.3)
    x[:, :5] *= 40
    hid = torch.randn(n, h, device=dev)
    heads = [torch.nn.Linear(h, g).to(dev) for _ in range(3)]
    with torch.no_grad():
        for i, L in enumerate(heads):
            L.weight.mul_(2.0 + i)
            L.bias.normal_(0, 1.0)
        heads[0].bias[9], heads[0].bias[10] = -60.0, 60.0   # beyond MeanAct's clamp on most rows
        heads[1].bias[11], heads[1].bias[12] = -60.0, 60.0
    sf = (torch.rand(n, device=dev, dtype=torch.float64) + 0.5) if with_sf else None
    return x, hid, heads, sf


@pytest.mark.parametrize("n,g,h,ridge,with_sf", [(64, 256, 16, 0.0, True), (257, 2000, 32, 0.5, True), (1000, 333, 8, 0.0, False), (130, 37, 8, 0.0, True),
                                                  (63, 1028, 8, 0.0, True)])
def test_zinb_heads_fused_equals_the_three_pass_form(cuda_device, n, g, h, ridge, with_sf):
    """dh_zinb_heads_fused_f32 (heads' loss + gradients over the raw outputs + bias column sums in one pass) against three HipLinear +
    zinb_nll_from_logits (dh_zinb_nll_logits_forward / _backward + dh_colsum_f32): the element arithmetic is the same code, so the
    raw-output gradients are BIT-identical for the same upstream constant; the loss differs by its summation order (the x = 0 terms: fp32
    groups of four there, float64 here: 1e-7);
    dW / db / dh by the order of the upstream scalar (applied to the small results instead of N x G elements: 1e-6).  Shapes: whole
    and partial 256-gene windows, a row count that is no multiple of 64, gene counts that rule the 16-byte path out (333, 37)."""
    from dance_amd import autograd, kernels
    torch.manual_seed(5)
    x, hid, heads, sf = _heads_case(n, g, h, cuda_device, with_sf)
    # the kernel alone on raw outputs
    raws = [kernels.gemm(hid, L.weight.detach(), trans_b=True, bias=L.bias.detach()) for L in heads]
    unit = 1.0 / (n * g)
    up = torch.tensor([unit], dtype=torch.float64, device=cuda_device)
    want_loss = kernels.zinb_nll_forward(x, *raws, sf, ridge, logits=True).sum()
    want = kernels.zinb_nll_backward(x, *raws, sf, ridge, up, logits=True)
    got = [r.clone() for r in raws]
    total, db = kernels.zinb_heads_fused_(x, *got, sf, ridge, unit)
    assert abs(float(total) - float(want_loss)) <= 1e-7 * abs(float(want_loss))  # the three-pass kernel adds its x = 0 terms in fp32 groups of four
    for a, b, nm in zip(got, want, ("mean", "disp", "pi")):
        assert torch.equal(a, b), (nm, float((a - b).abs().max()))
    want_db = torch.stack([w.double().sum(0) for w in want])
    assert float((db.double() - want_db).abs().max()) <= 2e-6 * float(want_db.abs().max())
    again = [r.clone() for r in raws]
    total2, db2 = kernels.zinb_heads_fused_(x, *again, sf, ridge, unit)
    assert torch.equal(total, total2) and torch.equal(db, db2) and all(torch.equal(a, b) for a, b in zip(got, again))  # run to run
    # the autograd function against the three-pass composition, with an upstream factor and a gradient for the hidden matrix
    hid_a = hid.clone().requires_grad_(True)
    loss_a = autograd.zinb_heads_loss(hid_a, heads, x, sf, ridge)
    assert loss_a.dtype == torch.float64
    params = [p for L in heads for p in (L.weight, L.bias)]
    ga = torch.autograd.grad(loss_a * 0.7, [hid_a, *params])
    hid_b = hid.clone().requires_grad_(True)
    outs = [autograd.linear(hid_b, L.weight, L.bias) for L in heads]
    loss_b = autograd.zinb_nll_from_logits(x, *outs, sf, ridge)
    gb = torch.autograd.grad(loss_b * 0.7, [hid_b, *params])
    assert abs(float(loss_a) - float(loss_b)) <= 1e-7 * abs(float(loss_b))
    for a, b in zip(ga, gb):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-6


def test_zinb_heads_fused_edge_cases(cuda_device):
    """Empty inputs return zeros without a launch; mismatched leading dimensions and CPU tensors are refused; the three-dimensional /
    non-fp32 fall-back of ``autograd.mix`` is the torch expression."""
    from dance_amd import _lib, autograd, kernels
    x = torch.zeros(0, 8, device=cuda_device)
    r = [torch.zeros(0, 8, device=cuda_device) for _ in range(3)]
    total, db = kernels.zinb_heads_fused_(x, *r, None, 0.0, 1.0)
    assert float(total) == 0.0 and db.shape == (3, 8) and float(db.abs().sum()) == 0.0
    x = torch.ones(4, 8, device=cuda_device)
    wide = torch.zeros(4, 16, device=cuda_device)
    with pytest.raises(ValueError):
        kernels.zinb_heads_fused_(x, wide[:, :8], torch.zeros(4, 8, device=cuda_device), torch.zeros(4, 8, device=cuda_device), None, 0.0, 1.0)
    with pytest.raises(_lib.DanceHipError):
        kernels.zinb_heads_fused_(x.cpu(), *(torch.zeros(4, 8) for _ in range(3)), None, 0.0, 1.0)
    a, b = torch.randn(3, 4, 5, device=cuda_device), torch.randn(3, 4, 5, device=cuda_device)
    assert torch.equal(autograd.mix(a, b, 0.25, 0.75), 0.25 * a + 0.75 * b)
    # one row, one gene; a single window with a ragged tail; every count non-zero; every count zero
    for n, g, dens in ((1, 1, 0.5), (70, 260, 1.0), (70, 260, 0.0)):
        xx = (torch.poisson(torch.rand(n, g, device=cuda_device) * 3) + 1.0) * (torch.rand(n, g, device=cuda_device) < dens)
        raws = [torch.randn(n, g, device=cuda_device) for _ in range(3)]
        unit = 1.0 / (n * g)
        up = torch.tensor([unit], dtype=torch.float64, device=cuda_device)
        want = kernels.zinb_nll_backward(xx, *raws, None, 0.0, up, logits=True)
        want_loss = kernels.zinb_nll_forward(xx, *raws, None, 0.0, logits=True).sum()
        got = [t.clone() for t in raws]
        total, db = kernels.zinb_heads_fused_(xx, *got, None, 0.0, unit)
        assert all(torch.equal(p, q) for p, q in zip(got, want)), (n, g, dens)
        assert abs(float(total) - float(want_loss)) <= 1e-7 * abs(float(want_loss)) + 1e-12


def test_mix_equals_the_torch_expression_bit_for_bit(cuda_device):
    """dh_axpby_f32 behind autograd.mix: (1 - sigma) * h + sigma * t of scdsc.py:454-459 in one pass == the three torch kernels (each
    product rounded, then the sum), gradients a * g / b * g; 16-byte path and the element path (odd width, a column slice)."""
    from dance_amd import autograd
    torch.manual_seed(3)
    for n, w, sl in ((1000, 256, None), (257, 33, None), (64, 40, slice(4, 37))):
        for sigma in (0.5, 0.3, 1.0):
            h = torch.randn(n, w, device=cuda_device)
            t = torch.randn(n, w, device=cuda_device)
            if sl is not None:
                h, t = h[:, sl], t[:, sl]
            h1, t1 = h.detach().clone().requires_grad_(True), t.detach().clone().requires_grad_(True)
            h2, t2 = h.detach().clone().requires_grad_(True), t.detach().clone().requires_grad_(True)
            if sl is not None:
                got = autograd.mix(h1 * 1.0, t1 * 1.0, 1 - sigma, sigma)
            else:
                got = autograd.mix(h1, t1, 1 - sigma, sigma)
            want = (1 - sigma) * h2 + sigma * t2
            assert torch.equal(got, want)
            g = torch.randn_like(want)
            got.backward(g)
            want.backward(g)
            assert torch.equal(h1.grad, h2.grad) and torch.equal(t1.grad, t2.grad)


def test_scdsc_fit_fused_heads_equals_unfused(cuda_device, tmp_path):
    """ScDSC.fit with ``fuse_zinb_heads`` on and off: the same soft assignments and the same last loss within fp32 rounding of the
    heads' gradients."""
    import numpy as np
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    import scipy.sparse as sp
    rng = np.random.default_rng(0)
    n, g, k = 600, 96, 6
    x_raw = rng.poisson(rng.random((n, g)) * 2.0).astype(np.float32)
    x = np.log1p(x_raw / np.maximum(x_raw.sum(1, keepdims=True), 1) * 1e3).astype(np.float32)
    nbr = rng.integers(0, n, (n, k))
    adj = sp.csr_matrix((np.full(n * k, 1.0 / k, np.float32), (np.repeat(np.arange(n), k), nbr.ravel())), shape=(n, n))
    y = rng.integers(0, 4, n)
    res = {}
    for fused in (True, False):
        torch.manual_seed(0)
        np.random.seed(0)
        m = ScDSC(pretrain_path=str(tmp_path / f"ae{int(fused)}.pt"), n_enc_1=32, n_enc_2=16, n_enc_3=16, n_dec_1=16, n_dec_2=16, n_dec_3=32, n_z1=16, n_z2=8, n_z3=4, n_clusters=4,
                  n_input=g, device=str(cuda_device))
        m.fuse_zinb_heads = fused
        m.fit((adj, x, x_raw, x_raw.sum(1).astype(np.float64) + 1.0), y, lr=1e-3, epochs=4, pt_epochs=2, pt_batch_size=64)
        res[fused] = (m.predict_proba(), float(m.last_loss))
    assert np.allclose(res[True][0], res[False][0], rtol=1e-4, atol=1e-6)
    assert abs(res[True][1] - res[False][1]) <= 1e-5 * abs(res[False][1])


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
