"""Captured training steps (``torch.cuda.CUDAGraph`` = hipGraph) for the launch-bound mini-batch loops (graph-sc batch 128, scDeepSort
batch 500: a step is a few dozen ~5 us kernels).

One process: block rebuild + forward + loss + backward + optimiser are ONE graph.  More than one process (data parallel over the seed
cells, ``sharding.allreduce_gradients`` per step): a collective cannot sit inside the graph of a step that also owns the optimiser
update it feeds, so the step is TWO graphs sharing a memory pool — [rebuild, forward, loss, backward] and [optimiser] — with the
gradient all-reduce issued eagerly on the same stream in between.  The gradients live in the first graph's pool, so both graphs and
the all-reduce see the same tensors on every replay.

The parameters' ``.grad`` tensors — allocated inside the capture, written by every replay — must remain THE gradients for as long as
the graph is replayed.  An eager step next to a captured one (the short last batch of an epoch) therefore zeroes and accumulates into
them in place (``optimizer.zero_grad(set_to_none=False)``) instead of replacing them: with ``set_to_none=True`` there, the next
epoch's replays died with a GPU memory access fault from 100k cells x batch 128 upwards (toy sizes never showed it; holding extra
references to the tensors — ``keep_alive`` / ``kept`` below, kept as a second line of defence — was not enough by itself, so the
mechanism is the allocator's treatment of the graph's pool once eager gradients of the same parameters exist, not a plain
use-after-free).  Regression test: tests/test_gpu_fullsize.py::test_captured_fits_survive_the_eager_last_batch.
``CapturedStep(params=...)`` additionally ENFORCES the invariant: every replay first re-attaches the capture-time gradient tensors to
parameters whose ``.grad`` was replaced in the meantime (tests/test_gpu_capture_invariant.py)."""
from typing import Callable, Optional

import torch


def kernel_copy_(dst: torch.Tensor, src: torch.Tensor) -> torch.Tensor:
    """``dst.copy_(src)`` as an elementwise KERNEL.  ``Tensor.copy_`` / ``clone`` of contiguous same-dtype tensors are hipMemcpyAsync
    calls, which a capture records as memcpy NODES (and ``zero_()`` / hipMemsetAsync as memset nodes).  On ROCm 7.2 / gfx950 such a
    node was replayed with the wrong extent and fill pattern once eager copies had run between two replays (round 5's fault hunt,
    profiles/r05_replay_fault.md), so code that runs under capture moves data with kernels only."""
    if dst.dtype != src.dtype:
        raise TypeError(f"kernel_copy_: dtypes differ ({dst.dtype} <- {src.dtype})")
    # a BIT copy (``add(x, 0)`` turns -0.0 into +0.0 and promotes bool): x | 0 for integers / bool, x * 1 for floats
    if src.dtype.is_floating_point or src.dtype.is_complex:
        return torch.mul(src, 1, out=dst)
    return torch.bitwise_or(src, False if src.dtype == torch.bool else 0, out=dst)


def kernel_clone(x: torch.Tensor) -> torch.Tensor:
    """``x.clone()`` as an elementwise kernel (see ``kernel_copy_``)."""
    if x.dtype.is_floating_point or x.dtype.is_complex:
        return torch.mul(x, 1)
    return torch.bitwise_or(x, False if x.dtype == torch.bool else 0)


class CapturedStep:

    def __init__(self, forward_backward: Callable[[], object], optimiser_step: Callable[[], None], device, *,
                 between: Optional[Callable[[], None]] = None, split: bool = False, warmup: int = 2,
                 keep_alive: Optional[Callable[[], list]] = None, params: Optional[list] = None):
        """``forward_backward()`` -> the step's static outputs (tensors that every replay refreshes); ``optimiser_step()`` applies the
        gradients; ``between()`` (split mode) runs eagerly between the two graphs — the gradient all-reduce.  The callables run
        ``warmup`` times on a side stream first (allocator, lazily created optimiser state), then are recorded.  The caller restores
        model / optimiser state afterwards if capturing must not count as training."""
        self.split, self.between = bool(split), between
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                forward_backward()
                if self.split and between is not None:
                    between()
                optimiser_step()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        self.graph_opt = None
        if not self.split:
            with torch.cuda.graph(self.graph):
                self.outputs = forward_backward()
                optimiser_step()
        else:
            with torch.cuda.graph(self.graph):
                self.outputs = forward_backward()
            self.graph_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_opt, pool=self.graph.pool()):
                optimiser_step()
        torch.cuda.synchronize(device)
        self.kept = list(keep_alive()) if keep_alive is not None else []
        # the invariant of the module docstring, enforced instead of assumed: ``params`` = the parameters whose ``.grad`` the graph
        # writes.  Any eager code that replaced a gradient tensor since (``zero_grad()`` with its default set_to_none=True, user code,
        # ``model.zero_grad()``) gets the capture-time tensor put back before the next replay — the eager gradient has been consumed by
        # then (its optimiser step ran), and in split mode the all-reduce between the two graphs must see the tensors the graph wrote.
        self._grads = [(p, p.grad) for p in (params or []) if p.grad is not None]

    def restore_gradients(self) -> int:
        """Re-attach the capture-time gradient tensors where eager code replaced them; returns how many were re-attached."""
        n = 0
        for p, g in self._grads:
            if p.grad is not g:
                p.grad = g
                n += 1
        return n

    def replay(self):
        self.restore_gradients()
        self.graph.replay()
        if self.split:
            if self.between is not None:
                self.between()
            self.graph_opt.replay()
        return self.outputs
