"""GPU: scHeteroNet (HetConv on the CSR SpMM, two-hop pattern from the device SpGEMM, energy propagation) against
tests/golden/scheteronet.npz — outputs of the reference's OWN HeteroNet / scHeteroNet classes (scheteronet.py:281-789),
AST-lifted and run on torch-CPU over stand-ins for torch_sparse / torch_geometric (tests/golden/make_golden.py)."""
import os
import types

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "scheteronet.npz")
DEV = "cuda"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _model(gold):
    from dance_amd.modules.single_modality.cell_type_annotation.scheteronet import scHeteroNet
    n, d, c, hid = (int(v) for v in gold["sh_dims"])
    m = scHeteroNet(d, c, torch.from_numpy(gold["sh_edge_index"]), n, hid, 2, 0.0, True, DEV, 100.0)
    sd = {k.split("::", 1)[1]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sh_sd0::")}
    assert sorted(sd) == sorted(m.state_dict())          # the reference's parameter names: its checkpoints load unchanged
    m.load_state_dict(sd)
    ds = types.SimpleNamespace(x=torch.from_numpy(gold["sh_x"]).to(DEV), edge_index=torch.from_numpy(gold["sh_edge_index"]).to(DEV),
                               y=torch.from_numpy(gold["sh_y"])[:, None].to(DEV), splits={"train": torch.arange(0, n, 2, device=DEV)},
                               node_idx=torch.arange(n, device=DEV))
    return m, ds


def test_init_adj_and_forward_vs_reference(cuda_device, gold):
    m, ds = _model(gold)
    enc = m.encoder
    assert rel_err(enc.adj_t.to_dense().cpu().numpy(), gold["sh_adj_t"]) < 1e-6
    assert np.array_equal(enc.adj_t2.to_dense().cpu().numpy() != 0, gold["sh_adj_t2"] != 0)   # bit-exact two-hop pattern
    assert rel_err(enc.adj_t2.to_dense().cpu().numpy(), gold["sh_adj_t2"]) < 1e-6
    m.eval()
    with torch.no_grad():
        h, mean, disp, pi = enc(ds.x, ds.edge_index, decoder=True)
        assert rel_err(h.cpu().numpy(), gold["sh_logits"]) < 1e-4
        assert rel_err(mean.cpu().numpy(), gold["sh_mean"]) < 1e-4 and rel_err(disp.cpu().numpy(), gold["sh_disp"]) < 1e-4
        assert rel_err(pi.cpu().numpy(), gold["sh_pi"]) < 1e-4
        e = torch.from_numpy(gold["sh_e"]).to(DEV)
        assert rel_err(m.propagation(e, ds.edge_index, 2, 0.5).cpu().numpy(), gold["sh_prop"]) < 1e-5
        assert rel_err(m.two_hop_propagation(e, ds.edge_index, 1, 0.3).cpu().numpy(), gold["sh_prop2"]) < 1e-5
        assert rel_err(m.detect(ds, ds.node_idx, DEV, 1.0, True, False, 2, 0.5).cpu().numpy(), gold["sh_detect"]) < 1e-4
    prob = m.predict_proba(ds)
    assert prob.shape == gold["sh_logits"].shape and torch.allclose(prob.sum(1), torch.ones(prob.shape[0]), atol=1e-5)
    assert np.array_equal(m.predict(ds).numpy(), gold["sh_logits"].argmax(1))


def test_fit_step_vs_reference(cuda_device, gold):
    m, ds = _model(gold)
    adata = types.SimpleNamespace(raw=types.SimpleNamespace(X=gold["sh_counts"]), obs={"size_factors": gold["sh_size_factors"]})
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    loss = m.fit(ds, ds, True, adata, 0.5, 0.0, 0.4, torch.nn.NLLLoss(), opt)
    assert abs(float(loss) - float(gold["sh_loss"])) < 1e-4 * abs(float(gold["sh_loss"]))
    for k in gold.files:
        if k.startswith("sh_sd1::") and "num_batches_tracked" not in k:
            got = m.state_dict()[k.split("::", 1)[1]].cpu().numpy()
            assert np.abs(got - gold[k]).max() < 2e-3 * max(1.0, np.abs(gold[k]).max()), k   # one Adam step of size lr = 1e-2
    # the contrastive term runs (random mask: no reference value to compare with).  use_zinb stays on: with use_zinb=False the
    # reference's loss_compute (:663-666) unpacks a single logits tensor into four names and cannot run
    loss2 = m.fit(ds, ds, True, adata, 0.5, 0.3, 0.4, torch.nn.NLLLoss(), opt)
    assert torch.isfinite(loss2)


@pytest.mark.parametrize("n,k,drop_diag", [(500, 6, False), (500, 6, True), (3000, 11, False), (64, 3, True)])
def test_two_hop_pattern_vs_scipy(cuda_device, n, k, drop_diag):
    """dh_csr_two_hop_*: ((A A) - A) > 0 on random kNN-like patterns with and without self loops, against scipy."""
    from dance_amd import kernels
    rng = np.random.default_rng(n + k)
    rows = np.repeat(np.arange(n), k)
    cols = np.concatenate([rng.choice(n, k, replace=False) for _ in range(n)])
    a = sp.csr_matrix((np.ones(rows.size, np.float32), (rows, cols)), shape=(n, n))
    if not drop_diag:
        a = ((a + sp.eye(n)) > 0).astype(np.float32).tocsr()     # self loops present, as HeteronetGraph produces them
    else:
        a.setdiag(0)
        a.eliminate_zeros()
    a.sort_indices()
    ref = (a @ a - a)
    ref = (ref > 0).astype(np.float32).tolil()
    if drop_diag:
        ref.setdiag(0)
    ref = ref.tocsr()
    ref.eliminate_zeros()
    ref.sort_indices()
    rp2, c2 = kernels.csr_two_hop(torch.from_numpy(a.indptr.astype(np.int32)).to(DEV), torch.from_numpy(a.indices.astype(np.int32)).to(DEV),
                                  drop_diag=drop_diag)
    assert np.array_equal(rp2.cpu().numpy(), ref.indptr.astype(np.int32)) and np.array_equal(c2.cpu().numpy(), ref.indices.astype(np.int32))
