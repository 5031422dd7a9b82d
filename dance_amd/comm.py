"""Python face of the ``dh_comm_*`` entry points (include/dance_hip.h): RCCL collectives of the destination-range sharded GCN
layer called straight through the C ABI, for consumers that do not run ``torch.distributed``.

``dance_amd.sharding`` exchanges through ``torch.distributed`` ("nccl" = the same RCCL) because the bench's launch contract hands
it an initialised process group; this class is the same transport without torch in the loop: rank 0 creates the unique id
(``Communicator.unique_id()``), ships its 128 bytes to the other ranks over any host channel, and every rank constructs
``Communicator(world, rank, id)``.  ``from_torch_distributed`` does that bootstrap over an existing process group.
"""
import ctypes
from typing import Optional, Sequence

import torch

from . import _lib
from .kernels import _dev, _ld, _stream

UNIQUE_ID_BYTES = 128


class Communicator:

    def __init__(self, world: int, rank: int, unique_id: bytes):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError(f"unique id must be {UNIQUE_ID_BYTES} bytes")
        self._lib = _lib.load()
        _lib.require_device()
        handle = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        _lib.check(self._lib.dh_comm_init(ctypes.byref(handle), int(world), int(rank), buf), "dh_comm_init")
        self._h = handle
        self.world, self.rank = int(world), int(rank)
        self._comm_stream: Optional[torch.cuda.Stream] = None

    @staticmethod
    def unique_id() -> bytes:
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        _lib.check(_lib.load().dh_comm_unique_id(buf), "dh_comm_unique_id")
        return buf.raw

    @classmethod
    def single(cls) -> "Communicator":
        return cls(1, 0, cls.unique_id())

    @classmethod
    def from_torch_distributed(cls, group=None) -> "Communicator":
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls(world, rank, box[0])

    def close(self):
        if getattr(self, "_h", None) is not None:
            self._lib.dh_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- collectives (asynchronous on the current stream) --------------------------------------------------------------------
    def allgather_rows(self, local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        rows, width = local.shape
        if out is None:
            out = torch.empty((rows * self.world, width), dtype=torch.float32, device=local.device)
        if not local.is_contiguous() or not out.is_contiguous():
            raise ValueError("allgather_rows wants contiguous buffers")
        _lib.check(self._lib.dh_comm_allgather_rows_f32(self._h, _dev(local, torch.float32, "local", 2), rows, width,
                                                        _dev(out, torch.float32, "out", 2), _stream()), "dh_comm_allgather_rows_f32")
        return out

    def allreduce_(self, buf: torch.Tensor) -> torch.Tensor:
        if not buf.is_contiguous():
            raise ValueError("allreduce_ wants a contiguous buffer")
        _lib.check(self._lib.dh_comm_allreduce_f32(self._h, _dev(buf, torch.float32, "buf"), buf.numel(), _stream()), "dh_comm_allreduce_f32")
        return buf

    def _counts(self, counts: Sequence[int]):
        if len(counts) != self.world:
            raise ValueError(f"need {self.world} per-peer row counts, got {len(counts)}")
        return (ctypes.c_int64 * self.world)(*[int(c) for c in counts])

    def halo_exchange(self, send: torch.Tensor, send_rows: Sequence[int], recv: torch.Tensor, recv_rows: Sequence[int]) -> torch.Tensor:
        """All-to-all-v of contiguous row blocks ordered by peer rank; ``send`` / ``recv`` are [sum(rows), width]."""
        width = send.shape[1] if send.numel() else recv.shape[1]
        _lib.check(self._lib.dh_comm_halo_exchange_f32(self._h, _dev(send, torch.float32, "send", 2), self._counts(send_rows),
                                                       _dev(recv, torch.float32, "recv", 2), self._counts(recv_rows), width, _stream()),
                   "dh_comm_halo_exchange_f32")
        return recv

    def halo_spmm(self, rowptr, col, val, operand: torch.Tensor, n_local: int, send_idx: torch.Tensor, send_rows: Sequence[int],
                  recv_rows: Sequence[int], interior: torch.Tensor, boundary: torch.Tensor, *, bias=None, act: int = 0,
                  send_relu_mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Y = act(A_local operand + bias) with the halo rows of ``operand`` ([n_local + n_halo, width]; the own rows must be in
        place) fetched from the peers while the interior rows are aggregated (dh_comm_halo_spmm_f32)."""
        n_halo = operand.shape[0] - n_local
        width = operand.shape[1]
        if out is None:
            out = torch.empty((n_local, width), dtype=torch.float32, device=operand.device)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=operand.device)
        send_buf = torch.empty((max(int(sum(send_rows)), 1), _ld(operand)), dtype=torch.float32, device=operand.device)  # rows travel at the operand's stride
        send_buf.record_stream(self._comm_stream)
        operand.record_stream(self._comm_stream)
        _lib.check(self._lib.dh_comm_halo_spmm_f32(
            self._h, n_local, n_halo, width, _dev(rowptr, torch.int32, "rowptr", 1), _dev(col, torch.int32, "col", 1),
            _dev(val, torch.float32, "val", 1), _dev(operand, torch.float32, "operand", 2), _ld(operand), _dev(send_idx, torch.int32, "send_idx", 1),
            self._counts(send_rows), self._counts(recv_rows), send_buf.data_ptr(), _dev(interior, torch.int32, "interior", 1), interior.numel(),
            _dev(boundary, torch.int32, "boundary", 1), boundary.numel(), _dev(out, torch.float32, "out", 2), _ld(out),
            _dev(bias, torch.float32, "bias", 1), act, None if send_relu_mask is None else send_relu_mask.data_ptr(), _stream(),
            self._comm_stream.cuda_stream), "dh_comm_halo_spmm_f32")
        return out
