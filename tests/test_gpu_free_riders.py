"""GPU: the (f)4 free riders through the HIP kernels (dh_gemm_f32, dh_spmm_csr_f32, dh_edge_softmax_shift_f32, dh_sddmm_csr_f32,
dh_edge_softmax_backward_f32) against the reference's own classes' outputs (tests/golden/free_riders.npz)."""
import pytest

import free_riders_checks as checks

pytestmark = pytest.mark.gpu


def test_gcn_family(cuda_device):
    checks.check_gcn_family(cuda_device)


def test_scgnn2_gat(cuda_device):
    checks.check_gat(cuda_device)


def test_graphsci_gnnmodel(cuda_device):
    checks.check_graphsci(cuda_device)
