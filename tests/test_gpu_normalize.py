"""Device normalize_total / log1p / scale against the numpy restatement of scanpy's algorithm (oracle/normalize.py)."""
import numpy as np
import pytest
import torch

from oracle import normalize as onorm

pytestmark = pytest.mark.gpu


def counts(n, f, seed, zero_rows=()):
    rng = np.random.default_rng(seed)
    X = rng.poisson(rng.gamma(0.6, 2.0, size=(1, f)), size=(n, f)).astype(np.float32)
    X[:, 3] = 0            # a constant gene (std 0)
    X[5, 7] = 5000         # one gene dominating one cell -> "highly expressed"
    for r in zero_rows:
        X[r] = 0
    return X


@pytest.mark.parametrize("n,f", [(257, 130), (1000, 33)])
@pytest.mark.parametrize("target", [None, 1.0, 1e4])
@pytest.mark.parametrize("exclude", [False, True])
def test_normalize_total(n, f, target, exclude):
    from dance_amd.transforms import normalize as dn
    X = counts(n, f, 0, zero_rows=(2, ))
    want = onorm.normalize_total(X, target, exclude_highly_expressed=exclude, max_fraction=0.05)
    got, _ = dn.normalize_total(torch.as_tensor(X).cuda(), target, exclude_highly_expressed=exclude, max_fraction=0.05)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-6, atol=0)  # float: row sums are f64-accumulated here, pairwise f32 in numpy


@pytest.mark.parametrize("base", [None, 2, 10])
def test_log1p(base):
    from dance_amd.transforms import normalize as dn
    X = counts(300, 70, 1)
    got = dn.log1p(torch.as_tensor(X).cuda(), base).cpu().numpy()
    np.testing.assert_allclose(got, onorm.log1p(X, base), rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize("zero_center,max_value", [(True, None), (True, 10.0), (False, None), (False, 3.0)])
def test_scale(zero_center, max_value):
    from dance_amd.transforms import normalize as dn
    X = onorm.log1p(onorm.normalize_total(counts(1500, 90, 2)))
    want, mean, std = onorm.scale(X, zero_center, max_value)
    got, gmean, gstd = dn.scale(torch.as_tensor(X).cuda(), zero_center, max_value)
    np.testing.assert_allclose(gmean.cpu().numpy(), mean, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(gstd.cpu().numpy(), std, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-6, atol=1e-6)


def test_pipeline_transforms_match_oracle():
    """NormalizeTotal -> Log1P -> Scale as Compose'd by the clustering pipelines (scdsc.py:122-127)."""
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms import Compose
    from dance_amd.transforms.normalize import Log1P, NormalizeTotal, NormalizeTotalLog1P, Scale
    X = counts(400, 60, 3)
    d = Data(AnnDataLite(X.copy()))
    Compose(NormalizeTotal(max_fraction=1.0), Log1P(), Scale(max_value=10))(d)
    want, mean, _ = onorm.scale(onorm.log1p(onorm.normalize_total(X)), True, 10)
    np.testing.assert_allclose(d.data.X, want, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(d.data.var["mean"].to_numpy(), mean, rtol=1e-6)
    d2 = Data(AnnDataLite(X.copy()))
    NormalizeTotalLog1P(max_fraction=0.05)(d2)
    np.testing.assert_allclose(d2.data.X, onorm.log1p(onorm.normalize_total(X, None, True, 0.05)), rtol=3e-6, atol=1e-7)


def test_bad_arguments_fail_loudly():
    from dance_amd import kernels
    with pytest.raises(Exception):
        kernels.rowsum_masked(torch.zeros(4, 4))  # CPU tensor: no CPU path
    x = torch.zeros(4, 4, device="cuda")
    with pytest.raises(TypeError):
        kernels.col_standardize(x, None, torch.ones(4, device="cuda"))  # statistics must be float64
