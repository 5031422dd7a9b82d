"""Tensor-level wrappers over the C ABI (include/dance_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function below hands raw device
pointers + the current stream to libdancehip.so.  All inputs must already live on the GPU — these
wrappers never copy to the host and never compute on the CPU.
"""
from typing import Optional, Tuple

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
             reduce: int = REDUCE_SUM, out: Optional[torch.Tensor] = None, tag: str = "spmm_csr_f32") -> torch.Tensor:
    """Y = act(rowscale * reduce_e(val[e] * colscale[col[e]] * Z[col[e]]) + bias); see dh_spmm_csr_f32."""
    lib = _lib_ready()
    n_rows = rowptr.numel() - 1
    width = Z.shape[1]
    n_cols = Z.shape[0] if n_cols is None else n_cols
    if out is None:
        out = torch.empty((n_rows, width), dtype=torch.float32, device=Z.device)
    _call(tag, lib.dh_spmm_csr_f32, n_rows, n_cols, width, _dev(rowptr, torch.int32, "rowptr", 1),
          _dev(col, torch.int32, "col", 1), _dev(val, torch.float32, "val", 1),
          _dev(rowscale, torch.float32, "rowscale", 1), _dev(colscale, torch.float32, "colscale", 1),
          _dev(Z, torch.float32, "Z", 2), _ld(Z), _dev(out, torch.float32, "out", 2), _ld(out),
          _dev(bias, torch.float32, "bias", 1), act, reduce, _stream())
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


def gemm(A: torch.Tensor, B: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False,
         out: Optional[torch.Tensor] = None, accumulate: bool = False, tag: Optional[str] = None) -> torch.Tensor:
    """C (+)= op(A) @ op(B) on the f32 matrix cores; see dh_gemm_f32."""
    lib = _lib_ready()
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
    ws_bytes = lib.dh_gemm_f32_workspace_bytes(M, N, K, int(trans_a), int(trans_b))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=A.device) if ws_bytes else None
    tag = tag or f"gemm_f32_{'t' if trans_a else 'n'}{'t' if trans_b else 'n'}"
    _call(tag, lib.dh_gemm_f32, M, N, K, int(trans_a), int(trans_b), _dev(A, torch.float32, "A", 2), _ld(A),
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


def colsum(X: torch.Tensor) -> torch.Tensor:
    """out[j] = sum_i X[i, j] (deterministic two-pass)."""
    lib = _lib_ready()
    out = torch.empty(X.shape[1], dtype=torch.float32, device=X.device)
    ws_bytes = lib.dh_colsum_f32_workspace_bytes(X.shape[0], X.shape[1])
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=X.device)
    _call("colsum_f32", lib.dh_colsum_f32, X.shape[0], X.shape[1], _dev(X, torch.float32, "X", 2), _ld(X),
          out.data_ptr(), ws.data_ptr(), ws_bytes, _stream())
    return out
