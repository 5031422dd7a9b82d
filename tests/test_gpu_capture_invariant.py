"""GPU: CapturedStep keeps the capture-time gradient tensors attached (the invariant "p.grad must remain the tensors the
graph writes" was documented but unenforced — any eager zero_grad() with the default set_to_none=True, or user code replacing a
gradient, broke it silently; in split mode the all-reduce between the two graphs would then reduce stale eager tensors)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("split", [False, True])
def test_replay_reattaches_replaced_gradients(cuda_device, split):
    from dance_amd.capture import CapturedStep
    torch.manual_seed(0)
    model = torch.nn.Linear(8, 4).to(cuda_device)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    x = torch.randn(16, 8, device=cuda_device)
    seen = []

    def fwd_bwd():
        loss = model(x).square().mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        return loss.detach()

    step = CapturedStep(fwd_bwd, opt.step, cuda_device, split=split, between=(lambda: seen.append([p.grad.data_ptr() for p in model.parameters()])) if split else None,
                        keep_alive=lambda: [p.grad for p in model.parameters()], params=list(model.parameters()))
    captured = [p.grad for p in model.parameters()]
    ptrs = [g.data_ptr() for g in captured]
    seen.clear()
    w0 = model.weight.detach().clone()
    step.replay()
    torch.cuda.synchronize()
    assert step.restore_gradients() == 0 and not torch.equal(w0, model.weight)
    # an eager step that REPLACES the gradients (the default zero_grad), as user code next to a captured fit might
    opt.zero_grad()                       # set_to_none=True
    model(x).sum().backward()
    assert all(p.grad is not g for p, g in zip(model.parameters(), captured))
    loss_before = float(model(x).detach().square().mean())
    step.replay()                         # re-attaches, then replays
    torch.cuda.synchronize()
    assert all(p.grad is g for p, g in zip(model.parameters(), captured))
    assert [p.grad.data_ptr() for p in model.parameters()] == ptrs
    if split:
        assert seen and all(s == ptrs for s in seen)   # the between() hook (the all-reduce) saw the graph's own tensors, every time
    assert float(model(x).detach().square().mean()) < loss_before  # the replayed step still trains


def test_static_block_rebuild_in_a_graph_survives_eager_copies(cuda_device):
    """Round 5's root cause of the round-4 replay faults, as a regression test: ``StaticCellBlock.rebuild()`` captured ALONE in a
    hipGraph and replayed 300 times with an eager seed copy before every replay.  While the exclusive scan zeroed ``out[0]`` with
    hipMemsetAsync, the capture held a 4-byte memset NODE, and ROCm 7.2 replayed it — once eager copies had run between two replays —
    as a fill of the whole row-pointer array with 0x80 bytes: negative offsets, a store 8 GB below the buffer, a GPU memory access
    fault by the second or third replay (profiles/r05_replay_fault.md).  The library now zeroes with kernels only (dh::zero_async)."""
    import numpy as np

    from dance_amd.cellgraph import StaticCellBlock
    from test_gpu_fullsize import _cellgene_graph
    n_cells, n_genes, batch = 20_000, 500, 128
    cg = _cellgene_graph(n_cells, n_genes, 60, 16, seed=3)
    blk = StaticCellBlock(cg, batch)
    side = torch.cuda.Stream(device=cuda_device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        blk.rebuild()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        blk.rebuild()
    gen = torch.Generator(device=cuda_device).manual_seed(0)
    rp = cg.rowptr.cpu().numpy().astype(np.int64)
    for it in range(300):
        seeds = torch.randint(n_genes, n_genes + n_cells, (batch, ), device=cuda_device, generator=gen)
        blk.seeds.copy_(seeds)             # the eager copy between two replays
        graph.replay()
        if it % 60 == 0 or it == 299:
            torch.cuda.synchronize()
            s = seeds.cpu().numpy()
            want = np.concatenate(([0], np.cumsum(rp[s + 1] - rp[s])))
            assert np.array_equal(blk.rowptr[:batch + 1].cpu().numpy(), want) and int(blk.bad) == 0
    torch.cuda.synchronize()
