import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, b):
    """max-norm relative error ||a-b||_inf / ||b||_inf (SURVEY.md §8c parity definition)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    denom = np.abs(b).max()
    return float(np.abs(a - b).max() / (denom if denom > 0 else 1.0))


@pytest.fixture(scope="session")
def golden_gcn():
    return dict(np.load(os.path.join(GOLDEN, "gcn_layers.npz")))


@pytest.fixture(scope="session")
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device("cuda:0")
