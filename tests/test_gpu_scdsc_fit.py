"""GPU: the ``ScDSC`` method wrapper (pre-training, joint loop, best-ARI checkpoint) against tests/golden/scdsc_fit.npz —
the reference's OWN ScDSC.fit (scdsc.py:200-288), AST-lifted and run on torch-CPU from the same initial weights and the
same torch seed (the only random draws are the shuffles of the pre-training DataLoader, taken from the CPU generator)."""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "scdsc_fit.npz")


def test_scdsc_fit_predict_vs_reference(cuda_device, tmp_path):
    from dance_amd.modules.single_modality.clustering.scdsc import ScDSC
    g = np.load(GOLD)
    kw = json.loads(str(g["sf_kw"]))
    n = g["sf_x"].shape[0]
    m = ScDSC(pretrain_path=str(tmp_path / "ae.pt"), device="cuda", **kw)
    sd0 = {k.split("::", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sf_sd0::")}
    assert sorted(sd0) == sorted(m.model.state_dict())       # reference checkpoints load unchanged
    m.model.load_state_dict(sd0)
    assert not any(p.requires_grad for p in m.model.ae.parameters())   # fix_module("model.ae") (:110)
    adj = sp.csr_matrix((g["sf_adj_data"], g["sf_adj_indices"], g["sf_adj_indptr"]), shape=(n, n))
    torch.manual_seed(10)
    m.fit((adj, g["sf_x"], g["sf_counts"], g["sf_n_counts"].astype(np.float64)), g["sf_y"], lr=1e-3, epochs=12, pt_epochs=3, pt_batch_size=32,
          pt_lr=1e-3)
    assert (tmp_path / "ae.pt").exists()                     # the pre-trained autoencoder is saved (base.py:104-106)
    q = m.predict_proba()
    assert q.shape == g["sf_q"].shape and np.allclose(q.sum(1), 1, atol=1e-5)
    assert rel_err(q, g["sf_q"]) < 5e-3                       # 12 pre-training + up to 12 joint Adam steps apart from fp32-CPU
    assert (m.predict() == g["sf_pred"]).mean() > 0.98
    for k in g.files:
        if k.startswith("sf_sd1::") and "num_batches_tracked" not in k:
            got = m.model.state_dict()[k.split("::", 1)[1]].cpu().numpy()
            # Linear biases that feed a BatchNorm have a mathematically ZERO gradient; Adam normalises the rounding noise that
            # is left into +-lr steps, so those entries random-walk by up to lr x steps = 1.2e-2 on either side
            assert np.abs(got - g[k]).max() < 1.5e-2 * max(1.0, np.abs(g[k]).max()), k
    assert m.score(None, g["sf_y"]) == pytest.approx(float(__import__("sklearn.metrics").metrics.adjusted_rand_score(g["sf_y"], g["sf_pred"])), abs=0.05)
