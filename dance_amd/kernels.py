"""Tensor-level wrappers over the C ABI (include/dance_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function below hands raw device
pointers + the current stream to libdancehip.so.  All inputs must already live on the GPU — these
wrappers never copy to the host and never compute on the CPU.
"""
from typing import Optional, Tuple

import os

import torch

from . import _lib

ACT_NONE, ACT_RELU = 0, 1
REDUCE_SUM, REDUCE_MEAN = 0, 1
METRIC_EUCLIDEAN, METRIC_PEARSON, METRIC_SPEARMAN = 0, 1, 2


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py's roofline leg).  While active, every wrapper below
    brackets its C-ABI call with events recorded on the stream the kernel is launched on."""
    active: Optional["KernelTimer"] = None

    def __init__(self):
        self.records = []  # (name, start_event, end_event)

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None

    def summary(self):
        """name -> (launches, mean ms).  Call after torch.cuda.synchronize()."""
        acc = {}
        for name, a, b in self.records:
            n, t = acc.get(name, (0, 0.0))
            acc[name] = (n + 1, t + a.elapsed_time(b))
        return {k: (n, t / n) for k, (n, t) in acc.items()}


def _call(tag: str, fn, *args):
    """Invoke one C-ABI launcher on the current stream; raise on a negative status."""
    timer = KernelTimer.active
    if timer is not None and torch.cuda.is_current_stream_capturing():
        timer = None  # events recorded into a hipGraph capture cannot be timed; replays are not visible per launch either
    if timer is not None:
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        status = fn(*args)
        b.record()
        timer.records.append((tag, a, b))
    else:
        status = fn(*args)
    _lib.check(status, fn.__name__)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: Optional[torch.Tensor], dtype, name: str, ndim: Optional[int] = None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.DanceHipError(f"{name} must be a GPU tensor (got {t.device}); dance_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name} must be {ndim}-d, got shape {tuple(t.shape)}")
    if t.dim() >= 1 and t.numel() > 0 and t.stride(-1) != 1:
        raise ValueError(f"{name} must be contiguous in its last dimension")
    if t.dim() == 1 and not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t.data_ptr()


def _ld(t: torch.Tensor) -> int:
    # leading dimension in elements of a row-major 2-d tensor (rows may be strided)
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def _lib_ready():
    lib = _lib.load()
    _lib.require_device()
    return lib


def spmm_csr(rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], Z: torch.Tensor, *,
             n_cols: Optional[int] = None, rowscale: Optional[torch.Tensor] = None,
             colscale: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
             reduce: int = REDUCE_SUM, out: Optional[torch.Tensor] = None, rows: Optional[torch.Tensor] = None,
             tag: str = "spmm_csr_f32") -> torch.Tensor:
    """Y = act(rowscale * reduce_e(val[e] * colscale[col[e]] * Z[col[e]]) + bias); see dh_spmm_csr_f32.
    ``rows`` (int32 row ids) restricts the launch to those rows of ``out`` (dh_spmm_csr_rows_f32); the others are left untouched."""
    lib = _lib_ready()
    n_rows = rowptr.numel() - 1
    width = Z.shape[1]
    n_cols = Z.shape[0] if n_cols is None else n_cols
    if out is None:
        out = torch.empty((n_rows, width), dtype=torch.float32, device=Z.device)
    if rows is not None:
        _call(tag, lib.dh_spmm_csr_rows_f32, rows.numel(), _dev(rows, torch.int32, "rows", 1), n_cols, width,
              _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1),
              _dev(rowscale, torch.float32, "rowscale", 1), _dev(colscale, torch.float32, "colscale", 1),
              _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(out, torch.float32, "out", 2), _ld(out),
              _dev(bias, torch.float32, "bias", 1), act, reduce, _stream())
        return out
    _call(tag, lib.dh_spmm_csr_f32, n_rows, n_cols, width, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1),
          _dev(rowscale, torch.float32, "rowscale", 1), _dev(colscale, torch.float32, "colscale", 1),
          _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(out, torch.float32, "out", 2), _ld(out),
          _dev(bias, torch.float32, "bias", 1), act, reduce, _stream())
    return out


def sddmm_csr(rowptr: torch.Tensor, col: torch.Tensor, U: torch.Tensor, V: torch.Tensor, *,
              scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[e] = scale[e] * <U[row(e)], V[col(e)]> per stored edge (dh_sddmm_csr_f32): the edge-value gradient of
    ``spmm_csr`` and sparse inner-product scores."""
    lib = _lib_ready()
    n_rows = rowptr.numel() - 1
    out = torch.empty(col.numel(), dtype=torch.float32, device=U.device)
    if col.numel() == 0:
        return out
    bf16 = U.dtype == torch.bfloat16
    fn, dt = (lib.dh_sddmm_csr_bf16, torch.bfloat16) if bf16 else (lib.dh_sddmm_csr_f32, torch.float32)
    _call("sddmm_csr_bf16" if bf16 else "sddmm_csr_f32", fn, n_rows, V.shape[0], U.shape[1], _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(scale, torch.float32, "scale", 1), _dev(U, dt, "U", 2), _ld(U),
          _dev(V, dt, "V", 2), _ld(V), out.data_ptr(), _stream())
    return out


def relu_mask_bytes(n_rows: int, width: int) -> int:
    """Size of the fused-ReLU sign mask for an [n_rows, width] layer output; 0 = fused path not applicable."""
    return int(_lib.load().dh_relu_mask_bytes(n_rows, width))


def spmm_csr_relu(rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], Z: torch.Tensor, *,
                  n_cols: Optional[int] = None, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                  out_mask: Optional[torch.Tensor] = None, in_mask: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, rows: Optional[torch.Tensor] = None,
                  slices: Optional[Tuple[int, int]] = None, resident: Optional[Tuple[int, int]] = None,
                  tag: str = "spmm_csr_f32") -> torch.Tensor:
    """dh_spmm_csr_relu_f32: forward records the ReLU sign mask (out_mask), backward applies it to the gathered
    rows (in_mask).  Masks are uint8 tensors of ``relu_mask_bytes`` bytes.  ``rows``: as in ``spmm_csr``.
    ``slices`` = (begin, end) in units of 128 columns: only those column slices of the layer are computed
    (dh_spmm_csr_relu_slices_f32; Z, out and the masks are the whole layer's).  ``resident`` = (workgroups, shape) with
    ``slices``: the fixed-footprint form (dh_spmm_csr_relu_slices_resident_f32) that co-schedules with a 128 x 128 GEMM."""
    lib = _lib_ready()
    n_rows = rowptr.numel() - 1
    width = Z.shape[1]
    n_cols = Z.shape[0] if n_cols is None else n_cols
    if out is None:
        out = torch.empty((n_rows, width), dtype=torch.float32, device=Z.device)
    if slices is not None and resident is not None:
        wgs, shape = int(resident[0]), int(resident[1])
        if not (0 < wgs < 65536 and shape in (0, 1)):
            raise ValueError(f"spmm_csr_relu: resident=(workgroups, shape) out of range: {resident!r}")
        _call(tag, lib.dh_spmm_csr_relu_slices_resident_f32, n_rows if rows is None else rows.numel(), _dev(rows, torch.int32, "rows", 1), n_cols,
              width, int(slices[0]), int(slices[1]), _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1),
              _dev(val, torch.float32, "val", 1), _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(out, torch.float32, "out", 2), _ld(out),
              _dev(bias, torch.float32, "bias", 1), act, None if out_mask is None else out_mask.data_ptr(),
              None if in_mask is None else in_mask.data_ptr(), wgs | (shape << 16), _stream())
        return out
    if slices is not None:
        _call(tag, lib.dh_spmm_csr_relu_slices_f32, n_rows if rows is None else rows.numel(), _dev(rows, torch.int32, "rows", 1), n_cols, width,
              int(slices[0]), int(slices[1]), _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1),
              _dev(val, torch.float32, "val", 1), _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(out, torch.float32, "out", 2), _ld(out),
              _dev(bias, torch.float32, "bias", 1), act, None if out_mask is None else out_mask.data_ptr(),
              None if in_mask is None else in_mask.data_ptr(), _stream())
        return out
    if rows is not None:
        _call(tag, lib.dh_spmm_csr_relu_rows_f32, rows.numel(), _dev(rows, torch.int32, "rows", 1), n_cols, width,
              _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1),
              _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(out, torch.float32, "out", 2), _ld(out), _dev(bias, torch.float32, "bias", 1),
              act, None if out_mask is None else out_mask.data_ptr(), None if in_mask is None else in_mask.data_ptr(), _stream())
        return out
    _call(tag, lib.dh_spmm_csr_relu_f32, n_rows, n_cols, width, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1), _dev(Z, torch.float32, "Z", 2), _ld(Z),
          _dev(out, torch.float32, "out", 2), _ld(out), _dev(bias, torch.float32, "bias", 1), act,
          None if out_mask is None else out_mask.data_ptr(), None if in_mask is None else in_mask.data_ptr(), _stream())
    return out


def student_t_supported(n_clusters: int, d: int) -> bool:
    """Shapes the fused Student-t head covers (dh_student_t_supported)."""
    return bool(_lib.load().dh_student_t_supported(int(n_clusters), int(d)))


def student_t_forward(Z: torch.Tensor, MU: torch.Tensor, a: float, eps: float, pw: float, scale: float) -> torch.Tensor:
    """q[i, j] of the DEC heads from the embedding Z [N, d] and the centres MU [C, d] (dh_student_t_forward_f32)."""
    lib = _lib_ready()
    q = torch.empty((Z.shape[0], MU.shape[0]), dtype=torch.float32, device=Z.device)
    _call("student_t_forward_f32", lib.dh_student_t_forward_f32, Z.shape[0], MU.shape[0], Z.shape[1], _dev(Z, torch.float32, "Z", 2), _ld(Z),
          _dev(MU, torch.float32, "MU", 2), float(a), float(eps), float(pw), float(scale), q.data_ptr(), _ld(q), _stream())
    return q


def student_t_backward(Z: torch.Tensor, MU: torch.Tensor, a: float, eps: float, pw: float, scale: float, G: torch.Tensor, *,
                       want_dz: bool = True) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
    """(dZ or None, dMU) from G = d loss / d q (dh_student_t_backward_f32)."""
    lib = _lib_ready()
    n, c, d = Z.shape[0], MU.shape[0], Z.shape[1]
    dz = torch.empty_like(Z) if want_dz else None
    dmu = torch.empty_like(MU)
    ws_bytes = lib.dh_student_t_backward_workspace_bytes(n, c, d)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=Z.device)
    _call("student_t_backward_f32", lib.dh_student_t_backward_f32, n, c, d, _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(MU, torch.float32, "MU", 2),
          float(a), float(eps), float(pw), float(scale), _dev(G, torch.float32, "G", 2), _ld(G), None if dz is None else dz.data_ptr(),
          0 if dz is None else _ld(dz), dmu.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
    return dz, dmu


def gather_rows(X: torch.Tensor, idx: torch.Tensor, *, relu_mask: Optional[torch.Tensor] = None,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i] = X[idx[i]] (optionally times the ReLU sign mask recorded for X's rows): dh_gather_rows_f32."""
    lib = _lib_ready()
    if out is None:
        out = torch.empty((idx.numel(), X.shape[1]), dtype=torch.float32, device=X.device)
    _call("gather_rows_f32", lib.dh_gather_rows_f32, idx.numel(), X.shape[1], _dev(idx, torch.int32, "idx", 1), _dev(X, torch.float32, "X", 2),
          _ld(X), None if relu_mask is None else relu_mask.data_ptr(), _dev(out, torch.float32, "out", 2), _ld(out), _stream())
    return out


def relu_mask_apply(X: torch.Tensor, relu_mask: torch.Tensor, *, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = X * [Y > 0] from the sign mask recorded by the forward SpMM (dh_relu_mask_apply_f32), one streaming pass."""
    lib = _lib_ready()
    if out is None:
        out = torch.empty_like(X)
    _call("relu_mask_apply_f32", lib.dh_relu_mask_apply_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          relu_mask.data_ptr(), _dev(out, torch.float32, "out", 2), _ld(out), _stream())
    return out


def csr_transpose(rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], n_rows: int,
                  n_cols: int) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], torch.Tensor]:
    """CSR of A^T (stable by input position); returns (rowptr_t, col_t, val_t, perm)."""
    lib = _lib_ready()
    nnz = col.numel()
    dev = rowptr.device
    rowptr_t = torch.empty(n_cols + 1, dtype=torch.int32, device=dev)
    col_t = torch.empty(nnz, dtype=torch.int32, device=dev)
    perm = torch.empty(nnz, dtype=torch.int32, device=dev)
    val_t = torch.empty(nnz, dtype=torch.float32, device=dev) if val is not None else None
    ws_bytes = lib.dh_csr_transpose_workspace_bytes(n_rows, n_cols, nnz)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    _call("csr_transpose", lib.dh_csr_transpose, n_rows, n_cols, nnz, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1), rowptr_t.data_ptr(),
          col_t.data_ptr(), None if val_t is None else val_t.data_ptr(), perm.data_ptr(), ws.data_ptr(), ws_bytes,
          _stream())
    return rowptr_t, col_t, val_t, perm


# How fp32 GEMMs run on the matrix cores: "x3" = split-bf16 kernel (dh_gemm_f32x3: fp32 operands/result, bf16 x 3 operand
# split, 6 partial products, fp32 accumulation; fp32-level accuracy), "exact" = v_mfma_f32_32x32x2_f32 (dh_gemm_f32, a
# bit-exact k-ordered fmaf chain).  Set with DANCE_AMD_GEMM or per call.
GEMM_MODE = os.environ.get("DANCE_AMD_GEMM", "exact")
GEMM_TILE_AUTO, GEMM_TILE_256, GEMM_TILE_128 = 0, 1, 2
GEMM_SMALL = os.environ.get("DANCE_AMD_GEMM_SMALL", "1") != "0"  # small products on dh_gemm_f32_small (A/B switch)
import contextvars  # noqa: E402

_mini_batch_depth = contextvars.ContextVar("dance_amd_mini_batch_products", default=0)  # per thread / task: another thread's gemm() is not switched


class mini_batch_products:
    """``with kernels.mini_batch_products():`` — inside, fp32 products with K <= 512 and M N <= 2^20 run on dh_gemm_f32_small (one round
    trip per 32 x 32 tile).  Opt-in by the mini-batch training loops (GraphSC.fit, ScDeepSort.fit) and NOT the default: the small kernel
    sums in another order than dh_gemm_f32, and the full-batch layers promise that a row's result does not depend on how many rows the
    call holds (a rank's rows of a sharded layer are bit-identical to the single-GPU layer's, tests/test_gpu_sharded_one_gpu.py)."""

    def __enter__(self):
        self._token = _mini_batch_depth.set(_mini_batch_depth.get() + 1)
        return self

    def __exit__(self, *exc):
        _mini_batch_depth.reset(self._token)
        return False


def gemm(A: torch.Tensor, B: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False,
         out: Optional[torch.Tensor] = None, accumulate: bool = False, tag: Optional[str] = None,
         mode: Optional[str] = None, tile: int = GEMM_TILE_AUTO, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """C (+)= op(A) @ op(B), fp32 in / fp32 out, on the matrix cores; see dh_gemm_f32x3 / dh_gemm_f32 and GEMM_MODE.
    ``tile`` (exact mode): macro-tile request of dh_gemm_f32_ex.  ``bias`` [N] / ``act``: nn.Linear's bias and a ReLU in the output
    tile's store (dh_gemm_f32_bias_act, exact mode; the x3 mode runs dh_bias_act_f32 behind the product)."""
    lib = _lib_ready()
    mode = mode or GEMM_MODE
    if mode not in ("x3", "exact"):
        raise ValueError(f"gemm: mode must be 'x3' or 'exact', got {mode!r}")
    M = A.shape[1] if trans_a else A.shape[0]
    K = A.shape[0] if trans_a else A.shape[1]
    Kb = B.shape[1] if trans_b else B.shape[0]
    N = B.shape[0] if trans_b else B.shape[1]
    if K != Kb:
        raise ValueError(f"gemm: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        if accumulate:
            raise ValueError("gemm: accumulate=True needs an `out` tensor")
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    tag = tag or f"gemm_f32_{'t' if trans_a else 'n'}{'t' if trans_b else 'n'}"
    if (mode == "exact" and GEMM_SMALL and _mini_batch_depth.get() > 0 and not accumulate and tile == GEMM_TILE_AUTO and 1 <= K <= 512
            and M * N <= (1 << 20) and M > 0 and N > 0):
        # the mini-batch steps' launch-bound products: one round trip per 32 x 32 tile instead of a K walk (dh_gemm_f32_small)
        _call(tag, lib.dh_gemm_f32_small, M, N, K, int(trans_a), int(trans_b), _dev(A, torch.float32, "A", 2), _ld(A), _dev(B, torch.float32, "B", 2),
              _ld(B), _dev(out, torch.float32, "out", 2), _ld(out), _dev(bias, torch.float32, "bias", 1), int(act), _stream())
        return out
    if bias is not None or act != ACT_NONE:
        if accumulate or tile != GEMM_TILE_AUTO:
            raise ValueError("gemm: bias / act go with a plain product (no accumulate, automatic tile)")
        if mode == "exact":
            ws_bytes = lib.dh_gemm_f32_workspace_bytes(M, N, K, int(trans_a), int(trans_b))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device) if ws_bytes else None
            _call(tag, lib.dh_gemm_f32_bias_act, M, N, K, int(trans_a), int(trans_b), _dev(A, torch.float32, "A", 2), _ld(A),
                  _dev(B, torch.float32, "B", 2), _ld(B), _dev(out, torch.float32, "out", 2), _ld(out), _dev(bias, torch.float32, "bias", 1), int(act),
                  None if ws is None else ws.data_ptr(), ws_bytes, _stream())
            return out
        gemm(A, B, trans_a=trans_a, trans_b=trans_b, out=out, tag=tag, mode=mode)
        return bias_act_(out, bias, act)
    if mode == "exact" and tile != GEMM_TILE_AUTO:
        ws_bytes = lib.dh_gemm_f32_ex_workspace_bytes(M, N, K, int(trans_a), int(trans_b), tile)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device) if ws_bytes else None
        _call(tag, lib.dh_gemm_f32_ex, M, N, K, int(trans_a), int(trans_b), _dev(A, torch.float32, "A", 2), _ld(A),
              _dev(B, torch.float32, "B", 2), _ld(B), _dev(out, torch.float32, "out", 2), _ld(out), int(accumulate),
              None if ws is None else ws.data_ptr(), ws_bytes, tile, _stream())
        return out
    size_fn, fn = (lib.dh_gemm_f32x3_workspace_bytes, lib.dh_gemm_f32x3) if mode == "x3" else (lib.dh_gemm_f32_workspace_bytes, lib.dh_gemm_f32)
    ws_bytes = size_fn(M, N, K, int(trans_a), int(trans_b))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device) if ws_bytes else None
    _call(tag, fn, M, N, K, int(trans_a), int(trans_b), _dev(A, torch.float32, "A", 2), _ld(A),
          _dev(B, torch.float32, "B", 2), _ld(B), _dev(out, torch.float32, "out", 2), _ld(out), int(accumulate),
          None if ws is None else ws.data_ptr(), ws_bytes, _stream())
    return out


def relu_backward(Y: torch.Tensor, dY: torch.Tensor) -> torch.Tensor:
    """G = dY * (Y > 0)."""
    lib = _lib_ready()
    G = torch.empty(Y.shape, dtype=torch.float32, device=Y.device)
    _call("relu_backward_f32", lib.dh_relu_backward_f32, Y.shape[0], Y.shape[1], _dev(Y, torch.float32, "Y", 2),
          _ld(Y), _dev(dY, torch.float32, "dY", 2), _ld(dY), G.data_ptr(), _ld(G), _stream())
    return G


def bias_act_(X: torch.Tensor, bias: Optional[torch.Tensor], act: int = ACT_NONE) -> torch.Tensor:
    """In place: X = act(X + bias)."""
    lib = _lib_ready()
    _call("bias_act_f32", lib.dh_bias_act_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          _dev(bias, torch.float32, "bias", 1), act, _stream())
    return X


def gaussian_kernel(D: torch.Tensor, l: float, *, want_out: bool = True, want_rowsum: bool = False):
    """exp(-D^2 / (2 l^2)) of a dense [n, m] matrix; returns (out | None, rowsum | None).
    A 1-d ``D`` (the value array of a kNN-truncated CSR of distances) is processed as a tall [len/256, 256] view plus a
    tail row, so the launch fills the chip instead of handing the whole array to one wavefront."""
    lib = _lib_ready()
    if D.dim() == 1:
        if want_rowsum:
            raise ValueError("gaussian_kernel: row sums are undefined for a flat value array")
        total = D.numel()
        out = torch.empty(total, dtype=torch.float32, device=D.device)
        base, obase, main = _dev(D, torch.float32, "D", 1), out.data_ptr(), (total // 256) * 256
        if main:
            _call("gaussian_kernel_f32", lib.dh_gaussian_kernel_f32, main // 256, 256, base, 256, float(l), obase, 256, None, _stream())
        if total > main:
            _call("gaussian_kernel_f32", lib.dh_gaussian_kernel_f32, 1, total - main, base + 4 * main, total - main, float(l),
                  obase + 4 * main, total - main, None, _stream())
        return out, None
    n, m = D.shape
    out = torch.empty((n, m), dtype=torch.float32, device=D.device) if want_out else None
    rs = torch.empty(n, dtype=torch.float32, device=D.device) if want_rowsum else None
    _call("gaussian_kernel_f32", lib.dh_gaussian_kernel_f32, n, m, _dev(D, torch.float32, "D", 2), _ld(D), float(l),
          None if out is None else out.data_ptr(), m, None if rs is None else rs.data_ptr(), _stream())
    return out, rs


def softplus_rowsum(X: torch.Tensor) -> torch.Tensor:
    """rowsum[r] = sum_c softplus(X[r, c]) (dh_softplus_rowsum_f32)."""
    lib = _lib_ready()
    out = torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
    _call("softplus_rowsum_f32", lib.dh_softplus_rowsum_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X), out.data_ptr(),
          _stream())
    return out


def sigmoid_scale(X: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """scale * sigmoid(X) with a device-resident scalar ``scale`` (dh_sigmoid_scale_f32)."""
    lib = _lib_ready()
    out = torch.empty(X.shape, dtype=torch.float32, device=X.device)
    _call("sigmoid_scale_f32", lib.dh_sigmoid_scale_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          _dev(scale.reshape(1), torch.float32, "scale", 1), out.data_ptr(), _ld(out), _stream())
    return out


def gram_sigmoid_supported(n: int, d: int) -> bool:
    return bool(_lib_ready().dh_gram_sigmoid_supported(int(n), int(d)))


def gcn_narrow_supported(in_features: int, out_features: int) -> bool:
    return bool(_lib_ready().dh_gcn_narrow_supported(int(in_features), int(out_features)))


def gcn_narrow_forward(rowptr, col, val, X, W, bias=None, act: int = ACT_NONE, *, n_cols: Optional[int] = None, want_agg: bool = True):
    """(Y, agg): Y = act((A X) W + bias) in one gather kernel (dh_gcn_narrow_forward_f32); agg [n_rows, 64] = the aggregated rows
    (zero padded, column 63 = 1), the operand of ``gcn_narrow_backward``."""
    lib = _lib_ready()
    n_rows = rowptr.numel() - 1
    f, h = W.shape
    y = torch.empty((n_rows, h), dtype=torch.float32, device=X.device)
    agg = torch.empty((n_rows, 64), dtype=torch.float32, device=X.device) if want_agg else None
    _call("gcn_narrow_forward_f32", lib.dh_gcn_narrow_forward_f32, n_rows, X.shape[0] if n_cols is None else n_cols, f, h,
          _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1),
          _dev(X, torch.float32, "X", 2), _ld(X), _dev(W, torch.float32, "W", 2), _ld(W), _dev(bias, torch.float32, "bias", 1), act,
          None if agg is None else agg.data_ptr(), y.data_ptr(), _ld(y), _stream())
    return y, agg


def gcn_narrow_backward(agg, dY, in_features: int, *, y_act: Optional[torch.Tensor] = None, want_bias: bool = True):
    """(dW [in, out], db [out] or None) = [agg | 1]^T (dY * [y_act > 0]) in one streaming pass (dh_gcn_narrow_backward_f32)."""
    lib = _lib_ready()
    n_rows, h = dY.shape
    dw = torch.empty((in_features, h), dtype=torch.float32, device=dY.device)
    db = torch.empty(h, dtype=torch.float32, device=dY.device) if want_bias else None
    ws_bytes = lib.dh_gcn_narrow_backward_workspace_bytes(n_rows)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dY.device)
    _call("gcn_narrow_backward_f32", lib.dh_gcn_narrow_backward_f32, n_rows, in_features, h, _dev(agg, torch.float32, "agg", 2),
          _dev(dY, torch.float32, "dY", 2), _ld(dY), _dev(y_act, torch.float32, "y_act", 2), 0 if y_act is None else _ld(y_act), dw.data_ptr(),
          _ld(dw), None if db is None else db.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
    return dw, db


def zinb_nll_forward(X, mean, disp, pi, scale_factor: Optional[torch.Tensor], ridge_lambda: float = 0.0, *, logits: bool = False) -> torch.Tensor:
    """rowloss [n] float64 of the zero-inflated negative-binomial NLL (dh_zinb_nll_forward_f32); the loss is its sum / (n g).
    ``logits=True``: mean / disp / pi are the decoder heads' RAW outputs and MeanAct / DispAct / sigmoid are applied inside the kernel
    (dh_zinb_nll_logits_forward_f32)."""
    lib = _lib_ready()
    n, g = X.shape
    rowloss = torch.empty(n, dtype=torch.float64, device=X.device)
    _call("zinb_nll_forward_f32", lib.dh_zinb_nll_logits_forward_f32 if logits else lib.dh_zinb_nll_forward_f32, n, g, _dev(X, torch.float32, "X", 2), _ld(X), _dev(mean, torch.float32, "mean", 2),
          _ld(mean), _dev(disp, torch.float32, "disp", 2), _ld(disp), _dev(pi, torch.float32, "pi", 2), _ld(pi),
          _dev(scale_factor, torch.float64, "scale_factor", 1), float(ridge_lambda), rowloss.data_ptr(), _stream())
    return rowloss


def zinb_nll_backward(X, mean, disp, pi, scale_factor: Optional[torch.Tensor], ridge_lambda: float, upstream: torch.Tensor, *,
                      logits: bool = False):
    """(d mean, d disp, d pi) fp32 of ``zinb_nll_forward`` times the float64 device scalar ``upstream`` (dh_zinb_nll_backward_f32;
    ``logits=True``: w.r.t. the heads' raw outputs, dh_zinb_nll_logits_backward_f32)."""
    lib = _lib_ready()
    n, g = X.shape
    dm, dd, dp = (torch.empty((n, g), dtype=torch.float32, device=X.device) for _ in range(3))
    _call("zinb_nll_backward_f32", lib.dh_zinb_nll_logits_backward_f32 if logits else lib.dh_zinb_nll_backward_f32, n, g, _dev(X, torch.float32, "X", 2), _ld(X), _dev(mean, torch.float32, "mean", 2),
          _ld(mean), _dev(disp, torch.float32, "disp", 2), _ld(disp), _dev(pi, torch.float32, "pi", 2), _ld(pi),
          _dev(scale_factor, torch.float64, "scale_factor", 1), float(ridge_lambda), _dev(upstream, torch.float64, "upstream"), dm.data_ptr(),
          dd.data_ptr(), dp.data_ptr(), g, _stream())
    return dm, dd, dp


def zinb_heads_fused_(X, mean_raw, disp_raw, pi_raw, scale_factor: Optional[torch.Tensor], ridge_lambda: float, unit: float):
    """One pass over the three heads' raw outputs (dh_zinb_heads_fused_f32): returns (sum of the element losses as a float64 0-dim tensor,
    d bias [3, G] for a unit upstream) and OVERWRITES ``mean_raw`` / ``disp_raw`` / ``pi_raw`` with ``unit`` x the gradients of that sum
    w.r.t. them.  The three matrices share one leading dimension (rows of one allocation or three equal ones)."""
    import ctypes
    lib = _lib_ready()
    n, g = X.shape
    if not (_ld(mean_raw) == _ld(disp_raw) == _ld(pi_raw)):
        raise ValueError("zinb_heads_fused_: the three head outputs must share one leading dimension")
    n_loss, n_rb = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.dh_zinb_heads_fused_partials(n, g, ctypes.byref(n_loss), ctypes.byref(n_rb)), "dh_zinb_heads_fused_partials")
    loss_p = torch.empty(max(n_loss.value, 1), dtype=torch.float64, device=X.device)
    col_p = torch.empty((max(n_rb.value, 1), 3 * g), dtype=torch.float32, device=X.device)
    _call("zinb_heads_fused_f32", lib.dh_zinb_heads_fused_f32, n, g, _dev(X, torch.float32, "X", 2), _ld(X), _dev(mean_raw, torch.float32, "mean_raw", 2),
          _dev(disp_raw, torch.float32, "disp_raw", 2), _dev(pi_raw, torch.float32, "pi_raw", 2), _ld(mean_raw),
          _dev(scale_factor, torch.float64, "scale_factor", 1), float(ridge_lambda), float(unit), loss_p.data_ptr(), col_p.data_ptr(), _stream())
    if n == 0 or g == 0:
        return torch.zeros((), dtype=torch.float64, device=X.device), torch.zeros((3, g), dtype=torch.float32, device=X.device)
    return loss_p[:n_loss.value].sum(), colsum(col_p[:n_rb.value]).view(3, g)


def softmax_xent_sum(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int = -100, want_grad: bool = True):
    """(loss, d) of ``CrossEntropyLoss(reduction="sum")``: loss a 0-dim fp32 tensor, d = softmax(logits) - onehot(labels) or None
    (dh_softmax_xent_sum_f32)."""
    lib = _lib_ready()
    n, c = logits.shape
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    d = torch.empty((n, c), dtype=torch.float32, device=logits.device) if want_grad else None
    ws_bytes = lib.dh_softmax_xent_sum_workspace_bytes(n, c)
    ws = torch.empty(max(ws_bytes, 4), dtype=torch.uint8, device=logits.device)
    _call("softmax_xent_sum_f32", lib.dh_softmax_xent_sum_f32, n, c, _dev(logits, torch.float32, "logits", 2), _ld(logits),
          _dev(labels, torch.int64, "labels", 1), int(ignore_index), loss.data_ptr(), None if d is None else d.data_ptr(), c, ws.data_ptr(), ws_bytes,
          _stream())
    return loss, d


GRAM_SOFTPLUS, GRAM_SIGMOID_SQ = 0, 1


def gram_pairwise(Z: torch.Tensor, mode: int = GRAM_SOFTPLUS) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rowloss, O) with rowloss[i] = sum_j f(<z_i, z_j>) and O[i] = sum_j f'(<z_i, z_j>) z_j over ALL pairs, without the n x n
    logits (dh_gram_pairwise_f32): f = softplus (GRAM_SOFTPLUS, graph-sc's decoder) or sigmoid^2 (GRAM_SIGMOID_SQ, scTAG's)."""
    lib = _lib_ready()
    n, d = Z.shape
    out = torch.empty((n, d), dtype=torch.float32, device=Z.device)
    rowloss = torch.empty(n, dtype=torch.float32, device=Z.device)
    ws_bytes = lib.dh_gram_sigmoid_workspace_bytes(n, d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=Z.device) if ws_bytes else None
    _call("gram_sigmoid_f32" if mode == GRAM_SOFTPLUS else "gram_sigmoid_sq_f32", lib.dh_gram_pairwise_f32, int(mode), n, d,
          _dev(Z, torch.float32, "Z", 2), _ld(Z), out.data_ptr(), _ld(out), rowloss.data_ptr(), None if ws is None else ws.data_ptr(), ws_bytes,
          _stream())
    return rowloss, out


def gram_pairwise_rect(Zr: torch.Tensor, Z: torch.Tensor, mode: int = GRAM_SOFTPLUS) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rowloss, O) for the rows of ``Zr`` against ALL rows of ``Z``: rowloss[i] = sum_j f(<zr_i, z_j>), O[i] = sum_j f'(<zr_i, z_j>) z_j
    (dh_gram_pairwise_rect_f32) — a rank's row block of the pairwise pass when the cells are sharded (sharding.py)."""
    lib = _lib_ready()
    (nr, d), n = Zr.shape, Z.shape[0]
    if Z.shape[1] != d:
        raise ValueError("gram_pairwise_rect: Zr and Z must have the same width")
    out = torch.empty((nr, d), dtype=torch.float32, device=Z.device)
    rowloss = torch.empty(nr, dtype=torch.float32, device=Z.device)
    ws_bytes = lib.dh_gram_pairwise_rect_workspace_bytes(nr, n, d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=Z.device) if ws_bytes else None
    _call("gram_pairwise_rect_f32", lib.dh_gram_pairwise_rect_f32, int(mode), nr, n, d, _dev(Zr, torch.float32, "Zr", 2), _ld(Zr),
          _dev(Z, torch.float32, "Z", 2), _ld(Z), out.data_ptr(), _ld(out), rowloss.data_ptr(), None if ws is None else ws.data_ptr(), ws_bytes,
          _stream())
    return rowloss, out


def gram_sigmoid(Z: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rowloss, O) with rowloss[i] = sum_j softplus(<z_i, z_j>) and O[i] = sum_j sigmoid(<z_i, z_j>) z_j: the dense part of
    graph-sc's inner-product decoder loss and (x2) its gradient, without the B x B logits (dh_gram_sigmoid_f32)."""
    return gram_pairwise(Z, GRAM_SOFTPLUS)


def gram_listed_forward(Z: torch.Tensor, us: torch.Tensor, vs: torch.Tensor, pos_weight: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """(xe, term) for the listed y = 1 entries (us[e], vs[e]) of graph-sc's decoder target: xe = <z_us, z_vs>,
    term = pos_weight * softplus(-xe) - softplus(xe) (dh_gram_listed_forward_f32)."""
    lib = _lib_ready()
    n, d = Z.shape
    e = us.numel()
    xe = torch.empty(e, dtype=torch.float32, device=Z.device)
    term = torch.empty(e, dtype=torch.float32, device=Z.device)
    _call("gram_listed_forward_f32", lib.dh_gram_listed_forward_f32, n, d, e, _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(us, torch.int32, "us", 1),
          _dev(vs, torch.int32, "vs", 1), float(pos_weight), xe.data_ptr(), term.data_ptr(), _stream())
    return xe, term


def gram_listed_backward(Z: torch.Tensor, O: torch.Tensor, us: torch.Tensor, vs: torch.Tensor, xe: torch.Tensor, pos_weight: float,
                         scale: torch.Tensor) -> torch.Tensor:
    """dZ = scale * (2 O + the listed entries' corrections), scale a device scalar (dh_gram_listed_backward_f32)."""
    lib = _lib_ready()
    n, d = Z.shape
    out = torch.empty((n, d), dtype=torch.float32, device=Z.device)
    _call("gram_listed_backward_f32", lib.dh_gram_listed_backward_f32, n, d, us.numel(), _dev(Z, torch.float32, "Z", 2), _ld(Z),
          _dev(O, torch.float32, "O", 2), _ld(O), _dev(us, torch.int32, "us", 1), _dev(vs, torch.int32, "vs", 1), _dev(xe, torch.float32, "xe", 1),
          float(pos_weight), _dev(scale.reshape(1), torch.float32, "scale", 1), out.data_ptr(), _ld(out), _stream())
    return out


def gram_diag_backward(Z: torch.Tensor, O: torch.Tensor, xe: torch.Tensor, pos_weight: float, scale: torch.Tensor) -> torch.Tensor:
    """``gram_listed_backward`` for the identity list (us = vs = arange(n)): one elementwise pass (dh_gram_diag_backward_f32)."""
    lib = _lib_ready()
    n, d = Z.shape
    out = torch.empty((n, d), dtype=torch.float32, device=Z.device)
    _call("gram_diag_backward_f32", lib.dh_gram_diag_backward_f32, n, d, _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(O, torch.float32, "O", 2), _ld(O),
          _dev(xe, torch.float32, "xe", 1), float(pos_weight), _dev(scale.reshape(1), torch.float32, "scale", 1), out.data_ptr(), _ld(out), _stream())
    return out


def axpby(a: float, X: torch.Tensor, b: float = 0.0, Y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``a * X + b * Y`` (``Y`` None: ``a * X``) in one pass, rounded like the three torch kernels of the expression (dh_axpby_f32)."""
    lib = _lib_ready()
    out = torch.empty(X.shape, dtype=torch.float32, device=X.device)
    _call("axpby_f32", lib.dh_axpby_f32, X.shape[0], X.shape[1], float(a), _dev(X, torch.float32, "X", 2), _ld(X), float(b),
          _dev(Y, torch.float32, "Y", 2), 0 if Y is None else _ld(Y), out.data_ptr(), _ld(out), _stream())
    return out


def colsum(X: torch.Tensor) -> torch.Tensor:
    """out[j] = sum_i X[i, j] (deterministic two-pass)."""
    lib = _lib_ready()
    out = torch.empty(X.shape[1], dtype=torch.float32, device=X.device)
    ws_bytes = lib.dh_colsum_f32_workspace_bytes(X.shape[0], X.shape[1])
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=X.device)
    _call("colsum_f32", lib.dh_colsum_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          out.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
    return out


# ---- bf16 storage path (config C3) ----------------------------------------------------------------------------
DTYPE_F32, DTYPE_BF16 = 0, 1
_BF16 = torch.bfloat16


def _out_dtype(dtype) -> int:
    if dtype == torch.float32:
        return DTYPE_F32
    if dtype == _BF16:
        return DTYPE_BF16
    raise TypeError(f"output dtype must be torch.float32 or torch.bfloat16, got {dtype}")


def spmm_csr_bf16(rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], Z: torch.Tensor, *,
                  n_cols: Optional[int] = None, rowscale: Optional[torch.Tensor] = None,
                  colscale: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
                  reduce: int = REDUCE_SUM, out_dtype=_BF16, tag: str = "spmm_csr_bf16") -> torch.Tensor:
    """dh_spmm_csr_bf16: bf16 gathered operand, fp32 accumulation, fp32 or bf16 output."""
    lib = _lib_ready()
    n_rows, width = rowptr.numel() - 1, Z.shape[1]
    n_cols = Z.shape[0] if n_cols is None else n_cols
    out = torch.empty((n_rows, width), dtype=out_dtype, device=Z.device)
    _call(tag, lib.dh_spmm_csr_bf16, n_rows, n_cols, width, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1),
          _dev(rowscale, torch.float32, "rowscale", 1), _dev(colscale, torch.float32, "colscale", 1),
          _dev(Z, _BF16, "Z", 2), _ld(Z), out.data_ptr(), _ld(out), _out_dtype(out_dtype),
          _dev(bias, torch.float32, "bias", 1), act, reduce, _stream())
    return out


def sage_aggregate_bf16(rowptr, col, w, src_cell_id, dst_cell_id, alpha, H, *, out_dtype=_BF16) -> torch.Tensor:
    """dh_sage_aggregate_bf16: AdaptiveSAGE weighted mean over bf16 node features."""
    lib = _lib_ready()
    n_dst, n_src, width = rowptr.numel() - 1, H.shape[0], H.shape[1]
    out = torch.empty((n_dst, width), dtype=out_dtype, device=H.device)
    _call("sage_aggregate_bf16", lib.dh_sage_aggregate_bf16, n_dst, n_src, width, alpha.numel() - 2,
          _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(w, torch.float32, "w", 1),
          _dev(src_cell_id, torch.int32, "src_cell_id", 1), _dev(dst_cell_id, torch.int32, "dst_cell_id", 1),
          _dev(alpha.reshape(-1), torch.float32, "alpha", 1), _dev(H, _BF16, "H", 2), _ld(H), out.data_ptr(),
          _ld(out), _out_dtype(out_dtype), _stream())
    return out


def gemm_bf16(A: torch.Tensor, B: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False,
              bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, out: Optional[torch.Tensor] = None,
              out_dtype=_BF16, accumulate: bool = False, tag: Optional[str] = None) -> torch.Tensor:
    """C (+)= act(op(A) @ op(B) + bias) on the bf16 matrix cores, fp32 accumulation; see dh_gemm_bf16."""
    lib = _lib_ready()
    M = A.shape[1] if trans_a else A.shape[0]
    K = A.shape[0] if trans_a else A.shape[1]
    Kb = B.shape[1] if trans_b else B.shape[0]
    N = B.shape[0] if trans_b else B.shape[1]
    if K != Kb:
        raise ValueError(f"gemm_bf16: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        if accumulate:
            raise ValueError("gemm_bf16: accumulate=True needs an `out` tensor")
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    # K-contiguous operands are read in place when their rows are 16-byte aligned; make them so (the size query assumes it)
    if not trans_a and K % 8 == 0 and (A.data_ptr() % 16 or _ld(A) % 8):
        A = A.contiguous()
    if trans_b and K % 8 == 0 and (B.data_ptr() % 16 or _ld(B) % 8):
        B = B.contiguous()
    ws_bytes = lib.dh_gemm_bf16_workspace_bytes(M, N, K, int(trans_a), int(trans_b))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device) if ws_bytes else None
    tag = tag or f"gemm_bf16_{'t' if trans_a else 'n'}{'t' if trans_b else 'n'}"
    _call(tag, lib.dh_gemm_bf16, M, N, K, int(trans_a), int(trans_b), _dev(A, _BF16, "A", 2), _ld(A),
          _dev(B, _BF16, "B", 2), _ld(B), _dev(out, out.dtype, "out", 2), _ld(out), _out_dtype(out.dtype),
          _dev(bias, torch.float32, "bias", 1), act, int(accumulate), None if ws is None else ws.data_ptr(), ws_bytes,
          _stream())
    return out


def relu_backward_bf16(Y: torch.Tensor, dY: torch.Tensor) -> torch.Tensor:
    """G = dY * (Y > 0) on bf16 tensors."""
    lib = _lib_ready()
    G = torch.empty(Y.shape, dtype=_BF16, device=Y.device)
    _call("relu_backward_bf16", lib.dh_relu_backward_bf16, Y.shape[0], Y.shape[1], _dev(Y, _BF16, "Y", 2), _ld(Y),
          _dev(dY, _BF16, "dY", 2), _ld(dY), G.data_ptr(), _ld(G), _stream())
    return G


def colsum_bf16(X: torch.Tensor) -> torch.Tensor:
    """out[j] = sum_i X[i, j] of a bf16 matrix, accumulated and returned in fp32."""
    lib = _lib_ready()
    out = torch.empty(X.shape[1], dtype=torch.float32, device=X.device)
    ws_bytes = lib.dh_colsum_f32_workspace_bytes(X.shape[0], X.shape[1])
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=X.device)
    _call("colsum_bf16", lib.dh_colsum_bf16, X.shape[0], X.shape[1], _dev(X, _BF16, "X", 2), _ld(X), out.data_ptr(),
          ws.data_ptr(), ws_bytes, _stream())
    return out


# ---- graph builders ----------------------------------------------------------------------------------------
def pairwise_distance(X: torch.Tensor, metric: int = METRIC_EUCLIDEAN) -> torch.Tensor:
    """Dense [n, n] f32 distance matrix (dh_pairwise_distance_f32); spearman ranks the rows on the device first."""
    lib = _lib_ready()
    n, d = X.shape
    if metric == METRIC_SPEARMAN:
        ranked = torch.empty((n, d), dtype=torch.float32, device=X.device)
        _call("rank_rows_f32", lib.dh_rank_rows_f32, n, d, _dev(X, torch.float32, "X", 2), _ld(X), ranked.data_ptr(),
              _ld(ranked), _stream())
        X = ranked
    out = torch.empty((n, n), dtype=torch.float32, device=X.device)
    _call("pairwise_distance_f32", lib.dh_pairwise_distance_f32, n, d, _dev(X, torch.float32, "X", 2), _ld(X),
          out.data_ptr(), _ld(out), metric, _stream())
    return out


KNN_AUTO, KNN_SCAN, KNN_FILTER, KNN_GRID = 0, 1, 2, 3


def knn(X: torch.Tensor, k: int, q_begin: int = 0, q_end: Optional[int] = None, *,
        algo: int = KNN_AUTO) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact kNN (self included), ordered by (distance, index); returns (idx int32 [nq,k], dist f32 [nq,k]).
    ``algo``: KNN_SCAN (vector-ALU scan), KNN_FILTER (matrix-core filter + exact re-rank), KNN_GRID (cell grid; d <= 3, k <= 32: what KNN_AUTO
    picks for spatial coordinates) or KNN_AUTO — same result, bit for bit."""
    lib = _lib_ready()
    n, d = X.shape
    q_end = n if q_end is None else q_end
    nq = q_end - q_begin
    idx = torch.empty((nq, k), dtype=torch.int32, device=X.device)
    dist = torch.empty((nq, k), dtype=torch.float32, device=X.device)
    ws_bytes = lib.dh_knn_bruteforce_f32_workspace_bytes(n, d, nq, k, algo)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=X.device)
    _call("knn_bruteforce_f32", lib.dh_knn_bruteforce_f32, n, d, _dev(X, torch.float32, "X", 2), _ld(X), q_begin,
          q_end, k, algo, idx.data_ptr(), dist.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
    return idx, dist


def exclusive_scan(counts: torch.Tensor) -> torch.Tensor:
    """int32 counts[n] -> int32 rowptr[n+1]."""
    lib = _lib_ready()
    n = counts.numel()
    out = torch.empty(n + 1, dtype=torch.int32, device=counts.device)
    ws_bytes = lib.dh_exclusive_scan_i32_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=counts.device)
    _call("exclusive_scan_i32", lib.dh_exclusive_scan_i32, n, _dev(counts, torch.int32, "counts", 1), out.data_ptr(),
          ws.data_ptr(), ws_bytes, _stream())
    return out


def umap_connectivities(knn_idx: torch.Tensor, knn_dist: torch.Tensor):
    """kNN list -> symmetric fuzzy-simplicial-set CSR (rowptr, col, val) + (sigma, rho); see umap.hip."""
    lib = _lib_ready()
    n, k = knn_idx.shape
    dev = knn_idx.device
    w = torch.empty((n, k), dtype=torch.float32, device=dev)
    sigma = torch.empty(n, dtype=torch.float32, device=dev)
    rho = torch.empty(n, dtype=torch.float32, device=dev)
    ws = torch.empty(8, dtype=torch.uint8, device=dev)
    ip, dp = _dev(knn_idx, torch.int32, "knn_idx", 2), _dev(knn_dist, torch.float32, "knn_dist", 2)
    if not (knn_idx.is_contiguous() and knn_dist.is_contiguous()):
        raise ValueError("knn_idx / knn_dist must be contiguous [n, k]")
    _call("umap_membership_f32", lib.dh_umap_membership_f32, n, k, ip, dp, w.data_ptr(), sigma.data_ptr(),
          rho.data_ptr(), ws.data_ptr(), 8, _stream())
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    _call("knn_row_nnz", lib.dh_knn_row_nnz, n, k, ip, w.data_ptr(), counts.data_ptr(), _stream())
    rp_w = exclusive_scan(counts)
    nnz_w = int(rp_w[-1])
    col_w = torch.empty(nnz_w, dtype=torch.int32, device=dev)
    val_w = torch.empty(nnz_w, dtype=torch.float32, device=dev)
    _call("knn_graph_to_csr", lib.dh_knn_graph_to_csr, n, k, ip, w.data_ptr(), rp_w.data_ptr(), col_w.data_ptr(),
          val_w.data_ptr(), _stream())
    rp_t, col_t, val_t, _ = csr_transpose(rp_w, col_w, val_w, n, n)
    _call("csr_union_count", lib.dh_csr_union_count, n, rp_w.data_ptr(), col_w.data_ptr(), rp_t.data_ptr(),
          col_t.data_ptr(), counts.data_ptr(), _stream())
    rp = exclusive_scan(counts)
    nnz = int(rp[-1])
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float32, device=dev)
    _call("csr_fuzzy_union_fill", lib.dh_csr_fuzzy_union_fill, n, rp_w.data_ptr(), col_w.data_ptr(), val_w.data_ptr(),
          rp_t.data_ptr(), col_t.data_ptr(), val_t.data_ptr(), rp.data_ptr(), col.data_ptr(), val.data_ptr(), _stream())
    return (rp, col, val), (sigma, rho)


def dense_to_csr(X: torch.Tensor):
    """(rowptr, col, val) of the non-zeros of a dense device matrix in row-major order (np.nonzero order): dh_dense_nnz_count_f32,
    dh_exclusive_scan_i32, dh_dense_to_csr_f32.  One host read (the total count, to size col / val)."""
    lib = _lib_ready()
    n, m = X.shape
    counts = torch.empty(n, dtype=torch.int32, device=X.device)
    _call("dense_nnz_count_f32", lib.dh_dense_nnz_count_f32, n, m, _dev(X, torch.float32, "X", 2), _ld(X), counts.data_ptr(), _stream())
    rowptr = exclusive_scan(counts)
    nnz = int(rowptr[-1])
    col = torch.empty(nnz, dtype=torch.int32, device=X.device)
    val = torch.empty(nnz, dtype=torch.float32, device=X.device)
    if nnz == 0:
        return rowptr, col, val
    _call("dense_to_csr_f32", lib.dh_dense_to_csr_f32, n, m, _dev(X, torch.float32, "X", 2), _ld(X), rowptr.data_ptr(), col.data_ptr(),
          val.data_ptr(), _stream())
    return rowptr, col, val


def csr_row_normalize(rowptr: torch.Tensor, val: torch.Tensor) -> torch.Tensor:
    """out[e] = deg(row) * val[e] / sum(val[row])."""
    lib = _lib_ready()
    out = torch.empty_like(val)
    _call("csr_row_normalize_f32", lib.dh_csr_row_normalize_f32, rowptr.numel() - 1,
          _dev(rowptr, torch.int32, "rowptr", 1), _dev(val, torch.float32, "val", 1), out.data_ptr(), _stream())
    return out


def rowsum_masked(X: torch.Tensor, colmask: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib_ready()
    out = torch.empty(X.shape[0], dtype=torch.float32, device=X.device)
    _call("rowsum_masked_f32", lib.dh_rowsum_masked_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          _dev(colmask, torch.uint8, "colmask", 1), out.data_ptr(), _stream())
    return out


def rowscale_log1p(X: torch.Tensor, divisor: Optional[torch.Tensor], *, log1p: bool, base: Optional[float] = None, inplace: bool = False) -> torch.Tensor:
    lib = _lib_ready()
    out = X if inplace else torch.empty_like(X)
    _call("rowscale_log1p_f32", lib.dh_rowscale_log1p_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          _dev(divisor, torch.float32, "divisor", 1), int(log1p), float(base or 0.0), out.data_ptr(), _ld(out), _stream())
    return out


def col_standardize(X: torch.Tensor, mean: Optional[torch.Tensor], std: torch.Tensor, max_value: Optional[float] = None, inplace: bool = False):
    """(X - mean) / std per column with float64 statistics (numpy's in-place arithmetic), clipped to +-max_value."""
    lib = _lib_ready()
    out = X if inplace else torch.empty_like(X)
    _call("col_standardize_f32", lib.dh_col_standardize_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          _dev(mean, torch.float64, "mean", 1), _dev(std, torch.float64, "std", 1), float(max_value or 0.0), out.data_ptr(), _ld(out), _stream())
    return out


def col_moments(X: torch.Tensor, rows_per_block: int = 512):
    """float64 column sums of X and of fl32(X * X): (sum [n_cols], sumsq [n_cols])."""
    lib = _lib_ready()
    n, f = X.shape
    rows_per_block = max(int(rows_per_block), -(-n // 65535))
    nb = max(-(-n // rows_per_block), 1)
    partial = torch.zeros(nb, 2, f, dtype=torch.float64, device=X.device)
    _call("col_moments_f32", lib.dh_col_moments_f32, n, f, _dev(X, torch.float32, "X", 2), _ld(X), rows_per_block, partial.data_ptr(), _stream())
    tot = partial.sum(0)
    return tot[0], tot[1]


def col_any_gt(X: torch.Tensor, thresh: torch.Tensor) -> torch.Tensor:
    lib = _lib_ready()
    flag = torch.empty(X.shape[1], dtype=torch.uint8, device=X.device)
    _call("col_any_gt_f32", lib.dh_col_any_gt_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          _dev(thresh, torch.float32, "thresh", 1), flag.data_ptr(), _stream())
    return flag


def spatial_gaussian_knn(X: torch.Tensor, k: int, l: float = 0.0):
    """dh_spatial_gaussian_knn: CSR (rowptr, col, val) of every spot's k nearest spots (self included), columns ascending;
    val = exp(-d^2 / (2 l^2)) for l > 0, the distance d for l <= 0."""
    lib = _lib_ready()
    n, d = X.shape
    dev = X.device
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(n * k, dtype=torch.int32, device=dev)
    val = torch.empty(n * k, dtype=torch.float32, device=dev)
    ws_bytes = lib.dh_spatial_gaussian_knn_workspace_bytes(n, d, k)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    _call("spatial_gaussian_knn", lib.dh_spatial_gaussian_knn, n, d, _dev(X, torch.float32, "X", 2), _ld(X), k, float(l), rowptr.data_ptr(),
          col.data_ptr(), val.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
    return rowptr, col, val


ATT_SIGMOID, ATT_LEAKY_RELU = 0, 1


def edge_softmax(rowptr, col, a_src, a_dst, *, act: int = ATT_SIGMOID, negative_slope: float = 0.2,
                 shift: Optional[torch.Tensor] = None) -> torch.Tensor:
    """att[e] = softmax over each row's in-edges of act(a_src[col[e]] + a_dst[row]) (dh_edge_softmax_f32).  ``shift``: a
    one-element device tensor subtracted before exp instead of the row maximum (dh_edge_softmax_shift_f32, scGNN2's GAT)."""
    lib = _lib_ready()
    att = torch.empty(col.numel(), dtype=torch.float32, device=col.device)
    _call("edge_softmax_f32", lib.dh_edge_softmax_shift_f32, rowptr.numel() - 1, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(a_src, torch.float32, "a_src", 1), _dev(a_dst, torch.float32, "a_dst", 1), act,
          float(negative_slope), _dev(shift, torch.float32, "shift"), att.data_ptr(), _stream())
    return att


def edge_softmax_backward(rowptr, col, a_src, a_dst, att, datt, *, act: int = ATT_SIGMOID, negative_slope: float = 0.2):
    """(dt [E], d_a_dst [n_rows]) of ``edge_softmax`` given d att (dh_edge_softmax_backward_f32)."""
    lib = _lib_ready()
    n_rows = rowptr.numel() - 1
    dt = torch.empty(col.numel(), dtype=torch.float32, device=col.device)
    d_dst = torch.empty(n_rows, dtype=torch.float32, device=col.device)
    _call("edge_softmax_backward_f32", lib.dh_edge_softmax_backward_f32, n_rows, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(a_src, torch.float32, "a_src", 1), _dev(a_dst, torch.float32, "a_dst", 1), act,
          float(negative_slope), _dev(att, torch.float32, "att", 1), _dev(datt, torch.float32, "datt", 1), dt.data_ptr(), d_dst.data_ptr(), _stream())
    return dt, d_dst


def csr_two_hop(rowptr: torch.Tensor, col: torch.Tensor, *, drop_diag: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pattern of ((A A) - A) > 0 for a square 0/1 CSR pattern (dh_csr_two_hop_*): returns (rowptr2, col2)."""
    lib = _lib_ready()
    n = rowptr.numel() - 1
    dev = rowptr.device
    rp, cp = _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1)
    rowcnt = torch.empty(n, dtype=torch.int32, device=dev)
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    _call("csr_two_hop_count", lib.dh_csr_two_hop_count, n, rp, cp, rowcnt.data_ptr(), overflow.data_ptr(), _stream())
    offs = exclusive_scan(rowcnt)
    total, ovf = int(offs[-1]), int(overflow)
    if ovf or total < 0:
        raise _lib.DanceHipError("csr_two_hop: more than 2^31 two-edge paths")
    flags = torch.empty(total, dtype=torch.int32, device=dev)
    ws_bytes = lib.dh_csr_two_hop_workspace_bytes(n, total)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    _call("csr_two_hop_expand", lib.dh_csr_two_hop_expand, n, total, rp, cp, offs.data_ptr(), int(drop_diag), flags.data_ptr(), ws.data_ptr(),
          ws_bytes, _stream())
    pos = exclusive_scan(flags)
    nnz2 = int(pos[-1])
    rowptr2 = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col2 = torch.empty(nnz2, dtype=torch.int32, device=dev)
    _call("csr_two_hop_compact", lib.dh_csr_two_hop_compact, n, total, flags.data_ptr(), pos.data_ptr(), rowptr2.data_ptr(), col2.data_ptr(),
          ws.data_ptr(), _stream())
    return rowptr2, col2


# ---- blocks of the full-neighbour sampler ---------------------------------------------------------------------
def block_build(rowptr: torch.Tensor, col: torch.Tensor, val: Optional[torch.Tensor], seeds: torch.Tensor, mark: torch.Tensor,
                lut: torch.Tensor):
    """dh_block_plan + dh_block_fill: the full-fan-out in-neighbour block of ``seeds`` (int64, unique).
    ``mark`` (uint8 [n_nodes], all zero — it is returned all zero) and ``lut`` (int32 [n_nodes]) persist with the graph.
    Returns (block_rowptr int32 [B+1], block_col int32 [E], block_val f32 [E] | None, src_ids int64 [B + others])."""
    lib = _lib_ready()
    n_nodes, n_seeds = rowptr.numel() - 1, seeds.numel()
    dev = rowptr.device
    brp = torch.empty(n_seeds + 1, dtype=torch.int32, device=dev)
    totals = torch.empty(2, dtype=torch.int32, device=dev)
    ws_bytes = lib.dh_block_workspace_bytes(n_nodes, n_seeds)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    sp = _dev(seeds, torch.int64, "seeds", 1)
    _call("block_plan", lib.dh_block_plan, n_nodes, n_seeds, sp, _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1),
          _dev(mark, torch.uint8, "mark", 1), _dev(lut, torch.int32, "lut", 1), brp.data_ptr(), totals.data_ptr(), ws.data_ptr(), ws_bytes,
          _stream())
    n_edges, n_others = totals.tolist()  # the one host round trip of a block
    bcol = torch.empty(n_edges, dtype=torch.int32, device=dev)
    bval = torch.empty(n_edges, dtype=torch.float32, device=dev) if val is not None else None
    src_ids = torch.empty(n_seeds + n_others, dtype=torch.int64, device=dev)
    _call("block_fill", lib.dh_block_fill, n_nodes, n_seeds, sp, rowptr.data_ptr(), col.data_ptr(), _dev(val, torch.float32, "val", 1),
          mark.data_ptr(), lut.data_ptr(), brp.data_ptr(), bcol.data_ptr(), None if bval is None else bval.data_ptr(), src_ids.data_ptr(),
          ws.data_ptr(), ws_bytes, _stream())
    return brp, bcol, bval, src_ids


def block_cells_static(rowptr, col, val, seeds: torch.Tensor, n_genes: int, brp: torch.Tensor, bcol: torch.Tensor, bval: torch.Tensor,
                       bad: torch.Tensor, ws: torch.Tensor):
    """dh_block_cells_static into caller-owned static buffers (brp int32 [B + 2], bcol int32 / bval f32 [E_max], bad int32 [1], ws
    uint8): the block of seed cells over the sources [seeds | all genes], no host round trip (hipGraph-capturable)."""
    lib = _lib_ready()
    _call("block_cells_static", lib.dh_block_cells_static, seeds.numel(), int(n_genes), bcol.numel(), _dev(seeds, torch.int64, "seeds", 1),
          _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1),
          _dev(brp, torch.int32, "brp", 1), _dev(bcol, torch.int32, "bcol", 1), _dev(bval, torch.float32, "bval", 1),
          _dev(bad, torch.int32, "bad", 1), ws.data_ptr(), ws.numel(), _stream())


DEGREE_BOTH, DEGREE_MEAN = 0, 1


def degree_scales(rowptr: torch.Tensor, col: torch.Tensor, n_rows: int, n_cols: int, mode: int = DEGREE_BOTH, *, n_pad: int = 0):
    """(rowscale [n_rows + n_pad], colscale [n_cols] or None): dh_csr_degree_scales_f32 over the entries of rows [0, n_rows) (``rowptr`` may
    be longer: a static block's padding row, whose scale — the last ``n_pad`` entries — is 1).  DEGREE_BOTH: D_in^-1/2, D_out^-1/2
    (degrees clamped at 1); DEGREE_MEAN: 1 / D_in."""
    lib = _lib_ready()
    dev = rowptr.device
    rowscale = torch.empty(n_rows + n_pad, dtype=torch.float32, device=dev)
    both = mode == DEGREE_BOTH
    colscale = torch.empty(n_cols, dtype=torch.float32, device=dev) if both else None
    count = torch.empty(n_cols, dtype=torch.int32, device=dev) if both else None
    if rowptr.numel() < n_rows + 1:
        raise ValueError("degree_scales: rowptr shorter than n_rows + 1")
    _call("csr_degree_scales_f32", lib.dh_csr_degree_scales_f32, int(n_rows), int(n_pad), int(n_cols), _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), int(mode), rowscale.data_ptr(), None if colscale is None else colscale.data_ptr(),
          None if count is None else count.data_ptr(), _stream())
    return rowscale, colscale


def block_cells_static_workspace_bytes(n_seeds: int) -> int:
    return int(_lib_ready().dh_block_cells_static_workspace_bytes(int(n_seeds)))


# ---- optimiser ---------------------------------------------------------------------------------------------
def adam_step(optimizer) -> bool:
    """One step of a ``torch.optim.Adam`` through dh_adam_step_f32 (two launches per 8 tensors; the state tensors the optimiser owns —
    exp_avg, exp_avg_sq, step — are updated in place, so checkpoints and ``state_dict`` stay what torch would have written).  Returns
    False without touching anything when the configuration is outside what the kernel covers (amsgrad / maximize / host-side step
    counters / non-fp32 or strided tensors / a state that the first ``optimizer.step()`` has not created yet): call ``optimizer.step()``."""
    import ctypes
    if type(optimizer) is not torch.optim.Adam:
        return False
    plans = []
    for grp in optimizer.param_groups:
        if grp.get("amsgrad") or grp.get("maximize") or grp.get("differentiable") or not isinstance(grp["lr"], (int, float)):
            return False
        ps = [p for p in grp["params"] if p.grad is not None]
        for p in ps:
            st = optimizer.state.get(p)
            if not st or "exp_avg" not in st or not torch.is_tensor(st.get("step")) or st["step"].device != p.device or st["step"].dtype != torch.float32:
                return False
            for t in (p, p.grad, st["exp_avg"], st["exp_avg_sq"]):
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != p.device.type:
                    return False
        if ps and ps[0].device.type == "cpu" and _cpu_adam is None:
            return False
        plans.append((grp, ps))
    lib = _lib_ready()
    for grp, ps in plans:
        if not ps:
            continue
        n = len(ps)
        tab = lambda f: (ctypes.c_void_p * n)(*[f(p) for p in ps])
        b1, b2 = grp["betas"]
        _call("adam_step_f32", lib.dh_adam_step_f32, n, tab(lambda p: p.data_ptr()), tab(lambda p: p.grad.data_ptr()),
              tab(lambda p: optimizer.state[p]["exp_avg"].data_ptr()), tab(lambda p: optimizer.state[p]["exp_avg_sq"].data_ptr()),
              tab(lambda p: optimizer.state[p]["step"].data_ptr()), (ctypes.c_int64 * n)(*[p.numel() for p in ps]), float(grp["lr"]), float(b1),
              float(b2), float(grp["eps"]), float(grp["weight_decay"]), _stream())
    return True


_cpu_adam = None  # (tests/cpu_ops.py patches adam_step itself; nothing here runs on CPU tensors)


# ---- DEC heads: target distribution and KL loss (dec_loss.hip) -----------------------------------------------------------------------
def dec_target(q: torch.Tensor, colsum_q: Optional[torch.Tensor] = None) -> torch.Tensor:
    """p = (q^2 / sum_i q) with rows normalised (spagcn.py:421-425) in one pass; ``colsum_q`` = the (all-reduced) column sums when given."""
    lib = _lib_ready()
    q = q if q.stride(-1) == 1 else q.contiguous()
    if colsum_q is None:
        colsum_q = colsum(q)
    p = torch.empty((q.shape[0], q.shape[1]), dtype=torch.float32, device=q.device)
    _call("dec_target_f32", lib.dh_dec_target_f32, q.shape[0], q.shape[1], _dev(q, torch.float32, "q", 2), _ld(q), _dev(colsum_q, torch.float32, "colsum_q", 1),
          p.data_ptr(), _ld(p), _stream())
    return p


def dec_kl_forward(p: torch.Tensor, q: torch.Tensor, eps: float, scale: float) -> torch.Tensor:
    """scale * sum_ij p log(p / (q + eps)) as a 0-d tensor (dh_dec_kl_forward_f32)."""
    lib = _lib_ready()
    out = torch.empty((), dtype=torch.float32, device=q.device)
    nb = int(lib.dh_dec_kl_workspace_bytes())
    ws = torch.empty(nb, dtype=torch.uint8, device=q.device)
    _call("dec_kl_forward_f32", lib.dh_dec_kl_forward_f32, q.shape[0], q.shape[1], _dev(p, torch.float32, "p", 2), _ld(p), _dev(q, torch.float32, "q", 2), _ld(q),
          float(eps), float(scale), out.data_ptr(), ws.data_ptr(), nb, _stream())
    return out


def dec_kl_backward(p: torch.Tensor, q: torch.Tensor, eps: float, scale: float, g: torch.Tensor) -> torch.Tensor:
    """dq = -g scale p / (q + eps) (dh_dec_kl_backward_f32); ``g``: the 0-d upstream gradient, on the device."""
    lib = _lib_ready()
    dq = torch.empty((q.shape[0], q.shape[1]), dtype=torch.float32, device=q.device)
    _call("dec_kl_backward_f32", lib.dh_dec_kl_backward_f32, q.shape[0], q.shape[1], _dev(p, torch.float32, "p", 2), _ld(p), _dev(q, torch.float32, "q", 2), _ld(q),
          float(eps), float(scale), g.data_ptr(), dq.data_ptr(), _ld(dq), _stream())
    return dq


# ---- AdaptiveSAGE ------------------------------------------------------------------------------------------
def sage_aggregate(rowptr, col, w, src_cell_id, dst_cell_id, alpha, H) -> torch.Tensor:
    """neigh[v] = mean_e alpha[idx(e)] * w_e * H[src(e)] (dh_sage_aggregate_f32)."""
    lib = _lib_ready()
    n_dst, n_src, width = rowptr.numel() - 1, H.shape[0], H.shape[1]
    out = torch.empty((n_dst, width), dtype=torch.float32, device=H.device)
    _call("sage_aggregate_f32", lib.dh_sage_aggregate_f32, n_dst, n_src, width, alpha.numel() - 2,
          _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(w, torch.float32, "w", 1),
          _dev(src_cell_id, torch.int32, "src_cell_id", 1), _dev(dst_cell_id, torch.int32, "dst_cell_id", 1),
          _dev(alpha.reshape(-1), torch.float32, "alpha", 1), _dev(H, torch.float32, "H", 2), _ld(H), out.data_ptr(),
          _ld(out), _stream())
    return out


def csr_densify_window(rowptr, col, val, col_begin: int, n_cols: int, *, rowscale=None, colscale=None, mean: bool = False,
                       dtype=torch.float32, ld: Optional[int] = None, max_row_nnz: int = 0) -> torch.Tensor:
    """dh_csr_densify_window: dense [n_rows, n_cols] copy (fp32 / bf16) of the columns [col_begin, col_begin + n_cols) of a
    CSR matrix, optionally scaled per row / per window column / by 1 / row degree.  ``ld`` >= n_cols pads the rows."""
    lib = _lib_ready()
    n_rows = rowptr.numel() - 1
    ld = n_cols if ld is None else ld
    out = torch.empty((n_rows, ld), dtype=dtype, device=rowptr.device)
    if ld > n_cols:
        out[:, n_cols:].zero_()
    _call("csr_densify_window", lib.dh_csr_densify_window, n_rows, max_row_nnz, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1), _dev(rowscale, torch.float32, "rowscale", 1),
          _dev(colscale, torch.float32, "colscale", 1), int(mean), col_begin, n_cols, out.data_ptr(), ld, _out_dtype(dtype), _stream())
    return out


def sage_tail(rowptr, col, w, src_cell_id, dst_cell_id, alpha, H, col_begin: int, n_cols: int, *, out_dtype=None) -> torch.Tensor:
    """dh_sage_tail: the mean-scaled AdaptiveSAGE contribution of the edges whose source lies outside the window."""
    lib = _lib_ready()
    n_dst, n_src, width = rowptr.numel() - 1, H.shape[0], H.shape[1]
    out_dtype = out_dtype or H.dtype
    out = torch.empty((n_dst, width), dtype=out_dtype, device=H.device)
    _call("sage_tail", lib.dh_sage_tail, n_dst, n_src, width, alpha.numel() - 2, col_begin, n_cols, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(w, torch.float32, "w", 1), _dev(src_cell_id, torch.int32, "src_cell_id", 1),
          _dev(dst_cell_id, torch.int32, "dst_cell_id", 1), _dev(alpha.reshape(-1), torch.float32, "alpha", 1), _dev(H, H.dtype, "H", 2),
          _ld(H), _out_dtype(H.dtype), out.data_ptr(), _ld(out), _out_dtype(out_dtype), _stream())
    return out


def sage_aggregate_dense(rowptr, col, w, src_cell_id, dst_cell_id, alpha, H, col_begin: int, n_cols: int, *,
                         dst_are_genes: bool = False, max_row_nnz: int = 0, out_dtype=None) -> torch.Tensor:
    """AdaptiveSAGE mean aggregation (the result of ``sage_aggregate``) through the matrix cores: the sources
    [col_begin, col_begin + n_cols) — the gene rows of H for cell destinations, the cell rows for gene destinations — enter as
    a dense weighted-adjacency operand of an MFMA GEMM (dh_csr_densify_window + dh_gemm_bf16 / dh_gemm_f32), every other
    in-edge (self loops) through dh_sage_tail.  bf16 H: the adjacency entries alpha * w / deg are rounded to bf16 (one more
    2^-9 rounding per term than the gather kernel); fp32 H: exact fp32 MFMA products, sums in a different order."""
    bf16 = H.dtype == _BF16
    out_dtype = out_dtype or H.dtype
    a = alpha.reshape(-1)
    if dst_are_genes:   # cell -> gene edge: alpha[cell_id of the destination gene] (gnn.py:74)
        rowscale, colscale = a[dst_cell_id.clamp(min=0).to(torch.int64)].contiguous(), None
    else:               # gene -> cell edge: alpha[cell_id of the source gene] (gnn.py:73)
        rowscale, colscale = None, a[src_cell_id[col_begin:col_begin + n_cols].clamp(min=0).to(torch.int64)].contiguous()
    k8 = (n_cols + 7) // 8 * 8 if bf16 else n_cols
    A = csr_densify_window(rowptr, col, w, col_begin, n_cols, rowscale=rowscale, colscale=colscale, mean=True,
                           dtype=H.dtype, ld=k8, max_row_nnz=max_row_nnz)
    Hw = H[col_begin:col_begin + n_cols]
    if k8 != n_cols:
        Hw = torch.cat((Hw, torch.zeros((k8 - n_cols, H.shape[1]), dtype=H.dtype, device=H.device)))
    out = sage_tail(rowptr, col, w, src_cell_id, dst_cell_id, alpha, H, col_begin, n_cols, out_dtype=out_dtype)
    if bf16:
        return gemm_bf16(A, Hw, out=out, accumulate=True, tag="gemm_bf16_sage_dense")
    if out_dtype != torch.float32:
        raise TypeError("fp32 features produce an fp32 result")
    return gemm(A, Hw, out=out, accumulate=True, tag="gemm_f32_sage_dense")


# How AdaptiveSAGE aggregates into CELL destinations when the gene rows form a known window of the sources (CellFeatureGraph
# layout, blocks of the device block builder): "mfma" = dh_sage_window_mfma where the shape fits, "gather" = always the gather
# kernels (dh_sage_aggregate_f32 / _bf16).  Set with DANCE_AMD_SAGE.
SAGE_MODE = os.environ.get("DANCE_AMD_SAGE", "mfma")
SAGE_BCM_MIN_ROWS = 256 * 128  # from here on dh_sage_window_mfma launches unsplit (>= 256 row blocks)


def sage_mfma_supported(n_cols: int, width: int, dtype) -> bool:
    return bool(_lib_ready().dh_sage_window_mfma_supported(int(n_cols), int(width), _out_dtype(dtype)))


def sage_aggregate_mfma(rowptr, col, w, src_cell_id, dst_cell_id, alpha, H, col_begin: int, n_cols: int, *, out_dtype=None, out=None) -> torch.Tensor:
    """AdaptiveSAGE mean aggregation for CELL destinations (the result of ``sage_aggregate``) with the gene window
    [col_begin, col_begin + n_cols) of H on the matrix cores and no dense adjacency in HBM (dh_sage_window_mfma: the
    workgroup densifies 128 cells x 128 genes at a time in LDS); the other in-edges (self loops, at the rows' ends) are
    added in the kernel's epilogue — one launch.
    fp32 H: entries and features enter as bf16 hi + lo pairs (three exact products per term, fp32 accumulation, ~1e-5
    worst-case relative error per term); bf16 H: features exact, entries hi + lo."""
    lib = _lib_ready()
    out_dtype = out_dtype or H.dtype
    a = alpha.reshape(-1)
    colscale = a[src_cell_id[col_begin:col_begin + n_cols].clamp(min=0).to(torch.int64)].contiguous()  # gnn.py:73
    n_dst = rowptr.numel() - 1
    out = _sage_out(out, n_dst, H, out_dtype)
    planned = (n_dst >= SAGE_BCM_MIN_ROWS or os.environ.get("DANCE_AMD_SAGE_MFMA", "") == "bcm") and os.environ.get("DANCE_AMD_SAGE_MFMA", "") != "v1"
    if planned and lib.dh_sage_window_mfma_planned_supported(n_dst, n_cols, H.shape[1], _out_dtype(H.dtype), H.data_ptr(), _ld(H), col.numel()):
        # unsplit launches (the full graph, large batches): the two-waves-per-SIMD kernel over a repacked ("block-chunk-major") copy of
        # the rows.  The repack depends on the graph only — kept per (rowptr, col, w) for as long as those tensors live unchanged
        plan = _sage_plan(rowptr, col, w, col_begin, n_cols)
        ws_bytes = lib.dh_sage_window_mfma_planned_workspace_bytes(n_cols, H.shape[1], _out_dtype(H.dtype))
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=H.device)
        _call("sage_window_mfma_planned", lib.dh_sage_window_mfma_planned, n_dst, H.shape[0], H.shape[1], col_begin, n_cols,
              _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(w, torch.float32, "w", 1),
              _dev(colscale, torch.float32, "colscale", 1), _dev(H, H.dtype, "H", 2), _ld(H), _out_dtype(H.dtype), out.data_ptr(), _ld(out),
              _out_dtype(out_dtype), col.numel(), _dev(src_cell_id, torch.int32, "src_cell_id", 1), _dev(dst_cell_id, torch.int32, "dst_cell_id", 1),
              _dev(a, torch.float32, "alpha", 1), a.numel() - 2, plan.data_ptr(), plan.numel(), ws.data_ptr(), ws_bytes, _stream())
        return out
    # (the larger size also holds the fp32 shares of a split launch: mini-batches spread over the chip instead of ceil(rows / 128) CUs)
    ws_bytes = lib.dh_sage_window_mfma_split_workspace_bytes(n_dst, n_cols, H.shape[1], _out_dtype(H.dtype))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=H.device)
    # src / dst ids + alpha given: the kernel adds the out-of-window edges (self loops) itself and writes the mean
    _call("sage_window_mfma", lib.dh_sage_window_mfma, n_dst, H.shape[0], H.shape[1], col_begin, n_cols,
          _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(w, torch.float32, "w", 1),
          _dev(colscale, torch.float32, "colscale", 1), _dev(H, H.dtype, "H", 2), _ld(H), _out_dtype(H.dtype), out.data_ptr(),
          _ld(out), _out_dtype(out_dtype), col.numel(), _dev(src_cell_id, torch.int32, "src_cell_id", 1),
          _dev(dst_cell_id, torch.int32, "dst_cell_id", 1), _dev(a, torch.float32, "alpha", 1), a.numel() - 2,
          ws.data_ptr(), ws_bytes, _stream())
    return out


def _sage_out(out, n_dst, H, out_dtype):
    """The result buffer of a window aggregation: fresh, or the caller's rows (a slice of a larger result: rows contiguous, any row stride)."""
    if out is None:
        return torch.empty((n_dst, H.shape[1]), dtype=out_dtype, device=H.device)
    if out.shape != (n_dst, H.shape[1]) or out.dtype != out_dtype or out.device != H.device or out.stride(1) != 1:
        raise ValueError(f"out must be a [{n_dst}, {H.shape[1]}] {out_dtype} tensor on {H.device} with contiguous rows")
    return out


_SAGE_PLANS = None  # graph.TensorKeyedCache: col tensor -> (extra key, plan bytes); dies with the tensor


def _sage_plan(rowptr, col, w, col_begin: int, n_cols: int, kind: str = "window") -> torch.Tensor:
    """dh_sage_window_plan (kind "window") or dh_sage_window_splitk_plan ("splitk") of (rowptr, col, w), cached by the identity +
    version of the three tensors (never by data_ptr)."""
    global _SAGE_PLANS
    from .graph import TensorKeyedCache
    import weakref
    if _SAGE_PLANS is None:
        _SAGE_PLANS = TensorKeyedCache()

    def ident(t):  # a view (block.rowptr_dst is a fresh slice on every access) is known by its base tensor + offset + length
        base = t._base if t._base is not None else t
        return base, (id(base), base._version, t.storage_offset(), t.numel())

    plans = _SAGE_PLANS.get(col)  # the plans made of this col tensor (dropped when it changes or dies): key -> (rowptr base, w base, plan)
    if plans is None:
        plans = _SAGE_PLANS.put(col, {})
    (rbase, rkey), (wbase, wkey) = ident(rowptr), ident(w)
    key = (kind, rkey, wkey, int(col_begin), int(n_cols))
    hit = plans.get(key)
    if hit is not None and hit[0]() is rbase and hit[1]() is wbase:
        return hit[2]
    lib = _lib_ready()
    n_dst = rowptr.numel() - 1
    size_fn, plan_fn = ((lib.dh_sage_window_plan_bytes, lib.dh_sage_window_plan) if kind == "window" else
                        (lib.dh_sage_window_splitk_plan_bytes, lib.dh_sage_window_splitk_plan))
    nbytes = size_fn(n_dst, n_cols, col.numel())
    if nbytes == 0:
        raise ValueError(f"sage plan ({kind}): {n_dst} rows x {n_cols} window columns, {col.numel()} entries not supported")
    plan = torch.empty(nbytes, dtype=torch.uint8, device=col.device)
    _call(f"sage_{kind}_plan", plan_fn, n_dst, col_begin, n_cols, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(w, torch.float32, "w", 1), col.numel(), plan.data_ptr(), nbytes, _stream())
    plans[key] = (weakref.ref(rbase), weakref.ref(wbase), plan)
    return plan


def sage_splitk_supported(n_dst: int, n_cols: int, width: int, dtype, nnz: int) -> bool:
    return bool(_lib_ready().dh_sage_window_splitk_supported(int(n_dst), int(n_cols), int(width), _out_dtype(dtype), None, 0, int(nnz)))


def sage_aggregate_splitk(rowptr, col, w, src_cell_id, dst_cell_id, alpha, H, col_begin: int, n_cols: int, *, out_dtype=None, out=None) -> torch.Tensor:
    """AdaptiveSAGE mean aggregation (the result of ``sage_aggregate``) for FEW destinations with very long rows — the gene nodes,
    ~1e5 cell in-neighbours each — whose CELL sources are the window [col_begin, col_begin + n_cols) of H (dh_sage_window_splitk: the
    window is the K dimension of the matrix-core loop, split over the chip; no dense adjacency).  The window's sources must all be
    cells (cell_id < 0): an in-window edge then carries alpha[cell_id of the destination] for a gene destination, the cell-cell
    alpha otherwise (gnn.py:72-76) — one scale per destination row."""
    lib = _lib_ready()
    out_dtype = out_dtype or H.dtype
    a = alpha.reshape(-1).float().contiguous()
    n_genes = a.numel() - 2
    did = dst_cell_id.to(torch.int64)
    rowscale = torch.where(did >= 0, a[did.clamp(min=0)], a[n_genes + 1]).contiguous()
    n_dst = rowptr.numel() - 1
    out = _sage_out(out, n_dst, H, out_dtype)
    plan = _sage_plan(rowptr, col, w, col_begin, n_cols, "splitk")
    ws_bytes = lib.dh_sage_window_splitk_workspace_bytes(n_dst, n_cols, H.shape[1], _out_dtype(H.dtype))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=H.device)
    _call("sage_window_splitk", lib.dh_sage_window_splitk, n_dst, H.shape[0], H.shape[1], col_begin, n_cols,
          _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(w, torch.float32, "w", 1),
          _dev(rowscale, torch.float32, "rowscale", 1), _dev(H, H.dtype, "H", 2), _ld(H), _out_dtype(H.dtype), out.data_ptr(), _ld(out),
          _out_dtype(out_dtype), col.numel(), _dev(src_cell_id, torch.int32, "src_cell_id", 1), _dev(dst_cell_id, torch.int32, "dst_cell_id", 1),
          _dev(a, torch.float32, "alpha", 1), n_genes, plan.data_ptr(), plan.numel(), ws.data_ptr(), ws_bytes, _stream())
    return out


def sage_alpha_grad(rowptr, col, w, src_cell_id, dst_cell_id, n_genes, H, dneigh) -> torch.Tensor:
    """dalpha[idx(e)] += w_e <H[src(e)], dneigh[dst(e)]> / deg(dst(e)) (dh_sage_alpha_grad_f32)."""
    lib = _lib_ready()
    n_dst, n_src, width = rowptr.numel() - 1, H.shape[0], H.shape[1]
    out = torch.empty(n_genes + 2, dtype=torch.float32, device=H.device)
    _call("sage_alpha_grad_f32", lib.dh_sage_alpha_grad_f32, n_dst, n_src, width, n_genes,
          _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1), _dev(w, torch.float32, "w", 1),
          _dev(src_cell_id, torch.int32, "src_cell_id", 1), _dev(dst_cell_id, torch.int32, "dst_cell_id", 1),
          _dev(H, torch.float32, "H", 2), _ld(H), _dev(dneigh, torch.float32, "dneigh", 2), _ld(dneigh),
          out.data_ptr(), _stream())
    return out


def cellgene_graph_assemble(rowptr_x, col_x, val_x, rowptr_t, col_t, val_t, perm_t, n_cells: int, n_genes: int):
    """CSR-by-destination CellFeatureGraph (+ reference edge ids) from X and X^T; see dh_cellgene_graph_assemble."""
    lib = _lib_ready()
    nnz = col_x.numel()
    dev = rowptr_x.device
    n_nodes, n_edges = n_cells + n_genes, 2 * nnz + n_cells + n_genes
    rowptr = torch.empty(n_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(n_edges, dtype=torch.int32, device=dev)
    val = torch.empty(n_edges, dtype=torch.float32, device=dev)
    eid = torch.empty(n_edges, dtype=torch.int32, device=dev)
    _call("cellgene_graph_assemble", lib.dh_cellgene_graph_assemble, n_cells, n_genes, nnz,
          _dev(rowptr_x, torch.int32, "rowptr_x", 1), _dev(col_x, torch.int32, "col_x", 1),
          _dev(val_x, torch.float32, "val_x", 1), _dev(rowptr_t, torch.int32, "rowptr_t", 1),
          _dev(col_t, torch.int32, "col_t", 1), _dev(val_t, torch.float32, "val_t", 1),
          _dev(perm_t, torch.int32, "perm_t", 1), rowptr.data_ptr(), col.data_ptr(), val.data_ptr(), eid.data_ptr(),
          _stream())
    return rowptr, col, val, eid
