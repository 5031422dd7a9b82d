"""GPU: GraphSCI on the HIP kernels — the loss terms and scores against the reference class's own numbers (tests/golden/graphsci.npz),
and the reference's preprocessing step list into a short fit / predict / score (tests/test_graphsci_model.py holds the bodies)."""
import pytest

import test_graphsci_model as tgm

pytestmark = pytest.mark.gpu


def test_graphsci_loss_and_scores_on_device(cuda_device, tmp_path, monkeypatch):
    tgm.check_graphsci_loss("cuda", tmp_path, monkeypatch)


def test_graphsci_pipeline_into_fit_on_device(cuda_device, tmp_path, monkeypatch):
    tgm.check_graphsci_pipeline("cuda", tmp_path, monkeypatch)
