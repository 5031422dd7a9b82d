"""Captured training steps (``torch.cuda.CUDAGraph`` = hipGraph) for the launch-bound mini-batch loops (graph-sc batch 128, scDeepSort
batch 500: a step is a few dozen ~5 us kernels).

One process: block rebuild + forward + loss + backward + optimiser are ONE graph.  More than one process (data parallel over the seed
cells, ``sharding.allreduce_gradients`` per step): a collective cannot sit inside the graph of a step that also owns the optimiser
update it feeds, so the step is TWO graphs sharing a memory pool — [rebuild, forward, loss, backward] and [optimiser] — with the
gradient all-reduce issued eagerly on the same stream in between.  The gradients live in the first graph's pool, so both graphs and
the all-reduce see the same tensors on every replay."""
from typing import Callable, Optional

import torch


class CapturedStep:

    def __init__(self, forward_backward: Callable[[], object], optimiser_step: Callable[[], None], device, *,
                 between: Optional[Callable[[], None]] = None, split: bool = False, warmup: int = 2):
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

    def replay(self):
        self.graph.replay()
        if self.split:
            if self.between is not None:
                self.between()
            self.graph_opt.replay()
        return self.outputs
