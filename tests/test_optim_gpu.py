"""graphpope_amd.optim.Adam (one launch per step) against torch.optim.Adam, the reference's optimiser (main.py:244).
Tolerance: 1e-6 relative to the largest magnitude after 6 steps (same float32 formulas, one fused rounding sequence)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from graphpope_amd import engine
    return engine.require_gpu()


def _close(a, b, rel=1e-6):
    scale = max(float(b.abs().max()), 1e-12)
    assert float((a - b).abs().max()) <= rel * scale, (float((a - b).abs().max()), scale)


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
def test_matches_torch_adam(weight_decay, dev):
    from graphpope_amd.optim import Adam
    torch.manual_seed(0)
    shapes = [(256, 756), (256,), (7, 256), (5000,), (3, 3, 3), (1,), (4097,)]
    ref_p = [torch.randn(s, device=dev).requires_grad_(True) for s in shapes]
    my_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    unused_r, unused_m = torch.randn(10, device=dev, requires_grad=True), torch.randn(10, device=dev, requires_grad=True)
    ref = torch.optim.Adam(ref_p + [unused_r], lr=0.01, weight_decay=weight_decay)
    mine = Adam(my_p + [unused_m], lr=0.01, weight_decay=weight_decay)
    for step in range(6):
        for a, b in zip(ref_p, my_p):
            g = torch.randn_like(a) * (10.0 ** (step - 3))
            a.grad, b.grad = g.clone(), g.clone()
        if step == 3:                                                   # ReduceLROnPlateau edits param_groups in place
            ref.param_groups[0]["lr"] = mine.param_groups[0]["lr"] = 0.001
        ref.step(); mine.step()
        for a, b in zip(ref_p, my_p):
            _close(b.detach(), a.detach())
    for a, b in zip(ref_p, my_p):
        _close(mine.state[b]["exp_avg"], ref.state[a]["exp_avg"])
        _close(mine.state[b]["exp_avg_sq"], ref.state[a]["exp_avg_sq"])
        assert mine.state[b]["step"] == int(ref.state[a]["step"]) == 6
    assert unused_m not in mine.state or not mine.state[unused_m]       # a parameter without a gradient is left alone


def test_state_dict_round_trips_with_torch(dev):
    from graphpope_amd.optim import Adam
    torch.manual_seed(1)
    a = torch.randn(300, device=dev, requires_grad=True)
    b = a.detach().clone().requires_grad_(True)
    mine, ref = Adam([a], lr=0.05), torch.optim.Adam([b], lr=0.05)
    for _ in range(3):
        g = torch.randn_like(a)
        a.grad, b.grad = g.clone(), g.clone()
        mine.step(); ref.step()
    sd = mine.state_dict()
    assert torch.is_tensor(sd["state"][0]["step"]) and float(sd["state"][0]["step"]) == 3.0
    ref2 = torch.optim.Adam([b], lr=0.05)
    ref2.load_state_dict(sd)                                            # torch accepts our checkpoint ...
    mine2 = Adam([a], lr=0.05)
    mine2.load_state_dict(ref.state_dict())                             # ... and we accept torch's
    g = torch.randn_like(a)
    a.grad, b.grad = g.clone(), g.clone()
    mine2.step(); ref2.step()
    _close(a.detach(), b.detach())
    torch.optim.lr_scheduler.ReduceLROnPlateau(mine2)                   # main.py:248 wraps the optimiser


def test_cpu_parameters_are_refused():
    from graphpope_amd.optim import Adam
    p = torch.zeros(4, requires_grad=True)
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        Adam([p]).step()
