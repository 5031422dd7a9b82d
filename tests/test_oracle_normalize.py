"""Pins of the normalisation oracle (oracle/normalize.py) on hand-computed vectors, and the host-only transform."""
import numpy as np

from oracle import normalize as onorm


def test_normalize_total_hand_vector():
    X = np.array([[1, 1, 2], [0, 3, 3], [0, 0, 0], [4, 0, 4]], dtype=np.float32)
    # counts 4, 6, 0, 8 -> median of the positive ones = 6; row 3 (all zero) divides by 1/6 and stays zero
    out = onorm.normalize_total(X)
    np.testing.assert_allclose(out.sum(1), [6, 6, 0, 6], rtol=1e-6)
    np.testing.assert_allclose(out[0], [1.5, 1.5, 3.0], rtol=1e-6)
    out = onorm.normalize_total(X, target_sum=1.0)
    np.testing.assert_allclose(out[3], [0.5, 0, 0.5], rtol=1e-6)


def test_normalize_total_excludes_highly_expressed():
    X = np.array([[90, 5, 5], [10, 10, 10], [1, 2, 7]], dtype=np.float32)
    # gene 0 is > 50% of cell 0, gene 2 is > 50% of cell 2  -> size factors from gene 1 only: 5, 10, 2 -> median 5
    out = onorm.normalize_total(X, exclude_highly_expressed=True, max_fraction=0.5)
    np.testing.assert_allclose(out[:, 1], [5, 5, 5], rtol=1e-6)
    np.testing.assert_allclose(out[0], [90, 5, 5], rtol=1e-6)
    np.testing.assert_allclose(out[2], [2.5, 5, 17.5], rtol=1e-6)


def test_log1p_and_base():
    X = np.array([[0, 1, 3]], dtype=np.float32)
    np.testing.assert_allclose(onorm.log1p(X), np.log([[1, 2, 4]]), rtol=1e-6)
    np.testing.assert_allclose(onorm.log1p(X, base=2), [[0, 1, 2]], rtol=1e-6)


def test_scale_hand_vector():
    X = np.array([[1, 5, 2], [3, 5, 2], [5, 5, 8]], dtype=np.float32)
    out, mean, std = onorm.scale(X)
    np.testing.assert_allclose(mean, [3, 5, 4])
    np.testing.assert_allclose(std, [2, 0, np.sqrt(12)])
    np.testing.assert_allclose(out[:, 0], [-1, 0, 1], atol=1e-7)
    np.testing.assert_allclose(out[:, 1], [0, 0, 0], atol=1e-7)    # constant gene: std 0 -> 1
    out, _, _ = onorm.scale(X, max_value=0.5)
    assert out.max() == 0.5 and out.min() == -0.5
    out, _, _ = onorm.scale(X, zero_center=False, max_value=1.0)
    assert out.max() == 1.0 and out.min() >= 0


def test_column_sum_normalize_host_transform():
    from dance_amd.data import AnnDataLite, Data
    from dance_amd.transforms.normalize import ColumnSumNormalize
    rng = np.random.default_rng(0)
    X = rng.random((12, 5)).astype(np.float32)
    d = Data(AnnDataLite(X.copy()), train_size=8, val_size=0, test_size=4)
    ColumnSumNormalize(split_names="ALL", mode="normalize", axis=0)(d)
    for idx in (d.train_idx, d.test_idx):
        np.testing.assert_allclose(d.data.X[idx].sum(0), 1.0, rtol=1e-5)
