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
