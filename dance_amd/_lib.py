"""ctypes binding of libdancehip.so (the C ABI declared in include/dance_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C dance_amd/csrc``).  There is no
CPU fallback: if the shared object is missing, or no HIP device is visible when a kernel is requested,
the call raises — a GPU box must never silently run something else.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DANCE_HIP_LIB", os.path.join(_HERE, "libdancehip.so"))  # override: A/B builds only

_lib = None


class DanceHipError(RuntimeError):
    """A libdancehip entry point returned a negative status."""


def _declare(lib):
    P = c_void_p  # every device pointer crosses the ABI as a raw address
    i64, i32 = c_int64, c_int
    sig = {
        "dh_version": (c_int, []),
        "dh_last_error_string": (c_char_p, []),
        "dh_device_count": (c_int, []),
        "dh_spmm_csr_f32": (c_int, [i64, i64, i64, P, P, P, P, P, P, i64, P, i64, P, i32, i32, P]),
        "dh_relu_mask_bytes": (c_size_t, [i64, i64]),
        "dh_spmm_csr_relu_f32": (c_int, [i64, i64, i64, P, P, P, P, i64, P, i64, P, i32, P, P, P]),
        "dh_spmm_csr_rows_f32": (c_int, [i64, P, i64, i64, P, P, P, P, P, P, i64, P, i64, P, i32, i32, P]),
        "dh_spmm_csr_relu_rows_f32": (c_int, [i64, P, i64, i64, P, P, P, P, i64, P, i64, P, i32, P, P, P]),
        "dh_spmm_csr_relu_slices_f32": (c_int, [i64, P, i64, i64, i64, i64, P, P, P, P, i64, P, i64, P, i32, P, P, P]),
        "dh_spmm_csr_relu_slices_resident_f32": (c_int, [i64, P, i64, i64, i64, i64, P, P, P, P, i64, P, i64, P, i32, P, P, i32, P]),
        "dh_student_t_supported": (c_int, [i64, i64]),
        "dh_student_t_forward_f32": (c_int, [i64, i64, i64, P, i64, P, c_float, c_float, c_float, c_float, P, i64, P]),
        "dh_student_t_backward_workspace_bytes": (c_size_t, [i64, i64, i64]),
        "dh_student_t_backward_f32": (c_int, [i64, i64, i64, P, i64, P, c_float, c_float, c_float, c_float, P, i64, P, i64, P, P, c_size_t, P]),
        "dh_relu_mask_apply_f32": (c_int, [i64, i64, P, i64, P, P, i64, P]),
        "dh_gather_rows_f32": (c_int, [i64, i64, P, P, i64, P, P, i64, P]),
        "dh_csr_transpose_workspace_bytes": (c_size_t, [i64, i64, i64]),
        "dh_csr_transpose": (c_int, [i64, i64, i64, P, P, P, P, P, P, P, P, c_size_t, P]),
        "dh_gemm_f32_workspace_bytes": (c_size_t, [i64, i64, i64, i32, i32]),
        "dh_gemm_f32": (c_int, [i64, i64, i64, i32, i32, P, i64, P, i64, P, i64, i32, P, c_size_t, P]),
        "dh_gemm_f32_ex_workspace_bytes": (c_size_t, [i64, i64, i64, i32, i32, i32]),
        "dh_gemm_f32_ex": (c_int, [i64, i64, i64, i32, i32, P, i64, P, i64, P, i64, i32, P, c_size_t, i32, P]),
        "dh_gemm_f32_bias_act": (c_int, [i64, i64, i64, i32, i32, P, i64, P, i64, P, i64, P, i32, P, c_size_t, P]),
        "dh_gemm_f32x3_workspace_bytes": (c_size_t, [i64, i64, i64, i32, i32]),
        "dh_gemm_f32x3": (c_int, [i64, i64, i64, i32, i32, P, i64, P, i64, P, i64, i32, P, c_size_t, P]),
        "dh_relu_backward_f32": (c_int, [i64, i64, P, i64, P, i64, P, i64, P]),
        "dh_bias_act_f32": (c_int, [i64, i64, P, i64, P, i32, P]),
        "dh_gaussian_kernel_f32": (c_int, [i64, i64, P, i64, c_double, P, i64, P, P]),
        "dh_axpby_f32": (c_int, [i64, i64, c_float, P, i64, c_float, P, i64, P, i64, P]),
        "dh_colsum_f32_workspace_bytes": (c_size_t, [i64, i64]),
        "dh_colsum_f32": (c_int, [i64, i64, P, i64, P, P, c_size_t, P]),
        "dh_pairwise_distance_f32": (c_int, [i64, i64, P, i64, P, i64, i32, P]),
        "dh_rank_rows_f32": (c_int, [i64, i64, P, i64, P, i64, P]),
        "dh_knn_bruteforce_f32_workspace_bytes": (c_size_t, [i64, i64, i64, i32, i32]),
        "dh_knn_bruteforce_f32": (c_int, [i64, i64, P, i64, i64, i64, i32, i32, P, P, P, c_size_t, P]),
        "dh_knn_filter_plan": (c_int, [i64, i64, i64, i32, P, i32]),
        "dh_umap_membership_f32": (c_int, [i64, i32, P, P, P, P, P, P, c_size_t, P]),
        "dh_knn_row_nnz": (c_int, [i64, i32, P, P, P, P]),
        "dh_knn_graph_to_csr": (c_int, [i64, i32, P, P, P, P, P, P]),
        "dh_csr_union_count": (c_int, [i64, P, P, P, P, P, P]),
        "dh_csr_fuzzy_union_fill": (c_int, [i64, P, P, P, P, P, P, P, P, P, P]),
        "dh_exclusive_scan_i32_workspace_bytes": (c_size_t, [i64]),
        "dh_exclusive_scan_i32": (c_int, [i64, P, P, P, c_size_t, P]),
        "dh_csr_row_normalize_f32": (c_int, [i64, P, P, P, P]),
        "dh_cellgene_graph_assemble": (c_int, [i64, i64, i64, P, P, P, P, P, P, P, P, P, P, P, P]),
        "dh_dense_nnz_count_f32": (c_int, [i64, i64, P, i64, P, P]),
        "dh_dense_to_csr_f32": (c_int, [i64, i64, P, i64, P, P, P, P]),
        "dh_sddmm_csr_f32": (c_int, [i64, i64, i64, P, P, P, P, i64, P, i64, P, P]),
        "dh_sddmm_csr_bf16": (c_int, [i64, i64, i64, P, P, P, P, i64, P, i64, P, P]),
        "dh_spmm_csr_bf16": (c_int, [i64, i64, i64, P, P, P, P, P, P, i64, P, i64, i32, P, i32, i32, P]),
        "dh_sage_aggregate_bf16": (c_int, [i64, i64, i64, i64, P, P, P, P, P, P, P, i64, P, i64, i32, P]),
        "dh_gemm_bf16_workspace_bytes": (c_size_t, [i64, i64, i64, i32, i32]),
        "dh_gemm_bf16": (c_int, [i64, i64, i64, i32, i32, P, i64, P, i64, P, i64, i32, P, i32, i32, P, c_size_t, P]),
        "dh_relu_backward_bf16": (c_int, [i64, i64, P, i64, P, i64, P, i64, P]),
        "dh_colsum_bf16": (c_int, [i64, i64, P, i64, P, P, c_size_t, P]),
        "dh_sage_aggregate_f32": (c_int, [i64, i64, i64, i64, P, P, P, P, P, P, P, i64, P, i64, P]),
        "dh_csr_densify_window": (c_int, [i64, i64, P, P, P, P, P, i32, i64, i64, P, i64, i32, P]),
        "dh_sage_window_mfma_supported": (c_int, [i64, i64, i32]),
        "dh_sage_window_mfma_workspace_bytes": (c_size_t, [i64, i64, i32]),
        "dh_sage_window_mfma_split_workspace_bytes": (c_size_t, [i64, i64, i64, i32]),
        "dh_sage_window_mfma_bcm_workspace_bytes": (c_size_t, [i64, i64, i64, i32, i64]),
        "dh_sage_window_plan_bytes": (c_size_t, [i64, i64, i64]),
        "dh_sage_window_plan": (c_int, [i64, i64, i64, P, P, P, i64, P, c_size_t, P]),
        "dh_sage_window_mfma_planned_supported": (c_int, [i64, i64, i64, i32, P, i64, i64]),
        "dh_sage_window_mfma_planned_workspace_bytes": (c_size_t, [i64, i64, i32]),
        "dh_sage_window_mfma_planned": (c_int, [i64, i64, i64, i64, i64, P, P, P, P, P, i64, i32, P, i64, i32, i64, P, P, P, i64, P, c_size_t, P, c_size_t, P]),
        "dh_sage_window_splitk_plan_bytes": (c_size_t, [i64, i64, i64]),
        "dh_sage_window_splitk_plan": (c_int, [i64, i64, i64, P, P, P, i64, P, c_size_t, P]),
        "dh_sage_window_splitk_supported": (c_int, [i64, i64, i64, i32, P, i64, i64]),
        "dh_sage_window_splitk_workspace_bytes": (c_size_t, [i64, i64, i64, i32]),
        "dh_sage_window_splitk": (c_int, [i64, i64, i64, i64, i64, P, P, P, P, P, i64, i32, P, i64, i32, i64, P, P, P, i64, P, c_size_t, P, c_size_t, P]),
        "dh_sage_window_mfma": (c_int, [i64, i64, i64, i64, i64, P, P, P, P, P, i64, i32, P, i64, i32, i64, P, P, P, i64, P, c_size_t, P]),
        "dh_sage_tail": (c_int, [i64, i64, i64, i64, i64, i64, P, P, P, P, P, P, P, i64, i32, P, i64, i32, P]),
        "dh_softplus_rowsum_f32": (c_int, [i64, i64, P, i64, P, P]),
        "dh_sigmoid_scale_f32": (c_int, [i64, i64, P, i64, P, P, i64, P]),
        "dh_gram_sigmoid_supported": (c_int, [i64, i64]),
        "dh_gram_sigmoid_workspace_bytes": (c_size_t, [i64, i64]),
        "dh_gram_sigmoid_f32": (c_int, [i64, i64, P, i64, P, i64, P, P, c_size_t, P]),
        "dh_gram_pairwise_f32": (c_int, [i32, i64, i64, P, i64, P, i64, P, P, c_size_t, P]),
        "dh_gram_pairwise_rect_workspace_bytes": (c_size_t, [i64, i64, i64]),
        "dh_gram_pairwise_rect_f32": (c_int, [i32, i64, i64, i64, P, i64, P, i64, P, i64, P, P, c_size_t, P]),
        "dh_gram_listed_forward_f32": (c_int, [i64, i64, i64, P, i64, P, P, c_float, P, P, P]),
        "dh_gram_listed_backward_f32": (c_int, [i64, i64, i64, P, i64, P, i64, P, P, P, c_float, P, P, i64, P]),
        "dh_gram_diag_backward_f32": (c_int, [i64, i64, P, i64, P, i64, P, c_float, P, P, i64, P]),
        "dh_rowsum_masked_f32": (c_int, [i64, i64, P, i64, P, P, P]),
        "dh_rowscale_log1p_f32": (c_int, [i64, i64, P, i64, P, i32, c_double, P, i64, P]),
        "dh_col_standardize_f32": (c_int, [i64, i64, P, i64, P, P, c_double, P, i64, P]),
        "dh_col_moments_f32": (c_int, [i64, i64, P, i64, i64, P, P]),
        "dh_col_any_gt_f32": (c_int, [i64, i64, P, i64, P, P, P]),
        "dh_spatial_gaussian_knn_workspace_bytes": (c_size_t, [i64, i64, i32]),
        "dh_spatial_gaussian_knn": (c_int, [i64, i64, P, i64, i32, c_double, P, P, P, P, c_size_t, P]),
        "dh_edge_softmax_f32": (c_int, [i64, P, P, P, P, i32, c_float, P, P]),
        "dh_edge_softmax_shift_f32": (c_int, [i64, P, P, P, P, i32, c_float, P, P, P]),
        "dh_edge_softmax_backward_f32": (c_int, [i64, P, P, P, P, i32, c_float, P, P, P, P, P]),
        "dh_csr_two_hop_count": (c_int, [i64, P, P, P, P, P]),
        "dh_csr_two_hop_workspace_bytes": (c_size_t, [i64, i64]),
        "dh_csr_two_hop_expand": (c_int, [i64, i64, P, P, P, i32, P, P, c_size_t, P]),
        "dh_csr_two_hop_compact": (c_int, [i64, i64, P, P, P, P, P, P]),
        "dh_block_workspace_bytes": (c_size_t, [i64, i64]),
        "dh_block_plan": (c_int, [i64, i64, P, P, P, P, P, P, P, P, c_size_t, P]),
        "dh_block_fill": (c_int, [i64, i64, P, P, P, P, P, P, P, P, P, P, P, c_size_t, P]),
        "dh_gcn_narrow_supported": (c_int, [i64, i64]),
        "dh_gcn_narrow_forward_f32": (c_int, [i64, i64, i64, i64, P, P, P, P, i64, P, i64, P, i32, P, P, i64, P]),
        "dh_gcn_narrow_backward_workspace_bytes": (c_size_t, [i64]),
        "dh_gcn_narrow_backward_f32": (c_int, [i64, i64, i64, P, P, i64, P, i64, P, i64, P, P, c_size_t, P]),
        "dh_zinb_nll_forward_f32": (c_int, [i64, i64, P, i64, P, i64, P, i64, P, i64, P, c_double, P, P]),
        "dh_zinb_nll_backward_f32": (c_int, [i64, i64, P, i64, P, i64, P, i64, P, i64, P, c_double, P, P, P, P, i64, P]),
        "dh_gemm_f32_small_supported": (c_int, [i64, i64, i64]),
        "dh_gemm_f32_small": (c_int, [i64, i64, i64, c_int, c_int, P, i64, P, i64, P, i64, P, c_int, P]),
        "dh_softmax_xent_sum_workspace_bytes": (c_size_t, [i64, i64]),
        "dh_softmax_xent_sum_f32": (c_int, [i64, i64, P, i64, P, i64, P, P, i64, P, c_size_t, P]),
        "dh_zinb_nll_logits_forward_f32": (c_int, [i64, i64, P, i64, P, i64, P, i64, P, i64, P, c_double, P, P]),
        "dh_zinb_nll_logits_backward_f32": (c_int, [i64, i64, P, i64, P, i64, P, i64, P, i64, P, c_double, P, P, P, P, i64, P]),
        "dh_zinb_heads_fused_partials": (c_int, [i64, i64, P, P]),
        "dh_zinb_heads_fused_f32": (c_int, [i64, i64, P, i64, P, P, P, i64, P, c_double, c_double, P, P, P]),
        "dh_comm_unique_id": (c_int, [P]),
        "dh_comm_init": (c_int, [P, i32, i32, P]),
        "dh_comm_destroy": (c_int, [P]),
        "dh_comm_world": (c_int, [P]),
        "dh_comm_rank": (c_int, [P]),
        "dh_comm_allgather_rows_f32": (c_int, [P, P, i64, i64, P, P]),
        "dh_comm_allreduce_f32": (c_int, [P, P, i64, P]),
        "dh_comm_halo_offsets": (c_int, [i32, i32, P, P, P, P, P, P]),
        "dh_comm_halo_exchange_f32": (c_int, [P, P, P, P, P, i64, P]),
        "dh_comm_halo_spmm_f32": (c_int, [P, i64, i64, i64, P, P, P, P, i64, P, P, P, P, P, i64, P, i64, P, i64, P, i32, P, P, P]),
        "dh_block_cells_static_workspace_bytes": (c_size_t, [i64]),
        "dh_block_cells_static": (c_int, [i64, i64, i64, P, P, P, P, P, P, P, P, P, c_size_t, P]),
        "dh_csr_degree_scales_f32": (c_int, [i64, i64, i64, P, P, i32, P, P, P, P]),
        "dh_adam_step_f32": (c_int, [i32, P, P, P, P, P, P, c_float, c_float, c_float, c_float, c_float, P]),
        "dh_sage_alpha_grad_f32": (c_int, [i64, i64, i64, i64, P, P, P, P, P, P, i64, P, i64, P, P]),
        "dh_dec_target_f32": (c_int, [i64, i64, P, i64, P, P, i64, P]),
        "dh_dec_kl_workspace_bytes": (c_size_t, []),
        "dh_dec_kl_forward_f32": (c_int, [i64, i64, P, i64, P, i64, c_float, c_double, P, P, c_size_t, P]),
        "dh_dec_kl_backward_f32": (c_int, [i64, i64, P, i64, P, i64, c_float, c_double, P, P, i64, P]),
        "dh_graphsc_step_supported": (c_int, [i64, i64, i64, i64]),
        "dh_graphsc_step_workspace_bytes": (c_size_t, [i64, i64, i64, i64, i64]),
        "dh_graphsc_steps": (c_int, [P, i64, i64, P]),            # dh_graphsc_step_t* (dance_amd/ministep.py mirrors the struct)
        "dh_scdeepsort_step_supported": (c_int, [i64, i64, i64, i64]),
        "dh_scdeepsort_step_workspace_bytes": (c_size_t, [i64, i64, i64, i64]),
        "dh_scdeepsort_steps": (c_int, [P, i64, i64, P]),         # dh_scdeepsort_step_t*
        "dh_ministep_dropout_mask_f32": (c_int, [i64, c_float, ctypes.c_uint64, ctypes.c_uint64, c_int32, P, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return sig


def load():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DanceHipError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(or `make -C dance_amd/csrc`) first; there is no CPU fallback")
        lib = ctypes.CDLL(LIB_PATH)
        lib._dh_signatures = _declare(lib)
        _lib = lib
    return _lib


def exported_symbols():
    """Names this binding expects (used by the header/library consistency test)."""
    return sorted(load()._dh_signatures)


def check(status: int, what: str = ""):
    if status != 0:
        msg = load().dh_last_error_string().decode("utf-8", "replace")
        raise DanceHipError(f"{what or 'libdancehip'} failed with status {status}: {msg}")


def require_device():
    """Raise unless a HIP device is visible (the product path never falls back to the CPU)."""
    if load().dh_device_count() < 1:
        raise DanceHipError("no HIP device visible: dance_amd kernels run on MI355X only (no CPU fallback)")
