"""Fused BatchNorm1d + ReLU + dropout epilogue (csrc/epilogue.hip) against torch's own modules.

The reference's hidden-layer tail is `x = self.bns[i](x); x = x.relu_(); x = F.dropout(x, p, training)`
(/root/reference/main.py:207-209): torch ops, installed here, so the parity is against the real thing.
Tolerances (fp32, different reduction order): outputs and running statistics 1e-5, gradients 1e-4, relative to
the largest magnitude.  Dropout draws its mask from a counter hash instead of torch's Philox stream, so the
p > 0 cases check the mask's properties and replay it through torch autograd.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from graphpope_amd import engine
    return engine.require_gpu()


def _close(got, want, rel):
    scale = max(float(want.detach().abs().max()), 1e-6)
    err = float((got.detach() - want.detach()).abs().max())
    assert err <= rel * scale, (err, scale)


def _pair(c, dev, seed):
    torch.manual_seed(seed)
    ref = torch.nn.BatchNorm1d(c).to(dev)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5)
        ref.bias.uniform_(-0.5, 0.5)
        ref.running_mean.uniform_(-1, 1)
        ref.running_var.uniform_(0.5, 2)
    mine = torch.nn.BatchNorm1d(c).to(dev)
    mine.load_state_dict(ref.state_dict())
    return ref, mine


@pytest.mark.parametrize("m,c", [(11264, 256), (1024, 256), (37, 10), (2, 5), (4099, 64), (513, 260)])
def test_training_mode_matches_torch(m, c, dev):
    from graphpope_amd.sage import bn_relu_dropout
    ref, mine = _pair(c, dev, m + c)
    x = (torch.randn(m, c, device=dev) * 2 + 0.7)
    xr, xm = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    gy = torch.randn(m, c, device=dev)
    yr = F.dropout(ref(xr).relu_(), p=0.0, training=True)
    ym = bn_relu_dropout(xm, mine, 0.0, True)
    _close(ym, yr, 1e-5)
    yr.backward(gy)
    ym.backward(gy)
    _close(xm.grad, xr.grad, 1e-4)
    _close(mine.weight.grad, ref.weight.grad, 1e-4)
    _close(mine.bias.grad, ref.bias.grad, 1e-4)
    _close(mine.running_mean, ref.running_mean, 1e-5)
    _close(mine.running_var, ref.running_var, 1e-5)
    assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1


def test_eval_mode_uses_running_statistics(dev):
    from graphpope_amd.sage import bn_relu_dropout
    ref, mine = _pair(256, dev, 3)
    ref.eval(); mine.eval()
    x = torch.randn(777, 256, device=dev)
    xr, xm = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yr = F.dropout(ref(xr).relu_(), p=0.5, training=False)
    ym = bn_relu_dropout(xm, mine, 0.5, False)
    _close(ym, yr, 1e-5)
    gy = torch.randn_like(x)
    yr.backward(gy); ym.backward(gy)
    _close(xm.grad, xr.grad, 1e-4)
    _close(mine.weight.grad, ref.weight.grad, 1e-4)
    _close(mine.bias.grad, ref.bias.grad, 1e-4)
    assert torch.equal(mine.running_mean, ref.running_mean) and int(mine.num_batches_tracked) == 0


@pytest.mark.parametrize("p", [0.5, 0.1, 0.9])
def test_dropout_mask_properties_and_replay(p, dev):
    from graphpope_amd.sage import bn_relu_dropout
    m, c = 11264, 256
    ref, mine = _pair(c, dev, 11)
    x = torch.randn(m, c, device=dev)
    xr, xm = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ym = bn_relu_dropout(xm, mine, p, True, seed=1234)
    act = ref(xr).relu()                                              # torch's BN + ReLU, no dropout
    positive = act > 0
    kept = (ym != 0) & positive
    frac = float(kept.sum()) / float(positive.sum())
    assert abs(frac - (1 - p)) < 0.004, frac                           # ~1.4 M draws: sigma < 5e-4
    # per column and per row the keep rate is the same (no structure in the hash)
    assert float((kept.sum(0) / positive.sum(0).clamp(min=1) - (1 - p)).abs().max()) < 0.03
    # kept elements are scaled by 1 / (1 - p), dropped ones are exactly zero
    yr = act * kept / (1 - p)                                          # replay the mask through torch autograd
    _close(ym, yr, 1e-5)
    gy = torch.randn_like(x)
    yr.backward(gy); ym.backward(gy)
    _close(xm.grad, xr.grad, 1e-4)
    _close(mine.weight.grad, ref.weight.grad, 1e-4)
    _close(mine.bias.grad, ref.bias.grad, 1e-4)
    # the mask is a pure function of the seed
    again = bn_relu_dropout(x, mine, p, True, seed=1234)
    other = bn_relu_dropout(x, mine, p, True, seed=1235)
    assert torch.equal(again != 0, ym != 0) and not torch.equal(other != 0, ym != 0)


def test_p_one_and_manual_seed(dev):
    from graphpope_amd.sage import bn_relu_dropout
    _, mine = _pair(64, dev, 1)
    x = torch.randn(300, 64, device=dev)
    assert float(bn_relu_dropout(x, mine, 1.0, True).abs().max()) == 0.0       # F.dropout(p=1) zeroes everything
    torch.manual_seed(7); a = bn_relu_dropout(x, mine, 0.5, True)
    torch.manual_seed(7); b = bn_relu_dropout(x, mine, 0.5, True)
    c = bn_relu_dropout(x, mine, 0.5, True)
    assert torch.equal(a != 0, b != 0) and not torch.equal(a != 0, c != 0)


def test_model_uses_the_fused_epilogue(dev, monkeypatch):
    """SAGE.forward routes the hidden-layer tail through the HIP op (a missing library would raise, not fall back)."""
    from graphpope_amd import sage
    calls = []
    real = sage._BnReluDropoutFn.apply
    monkeypatch.setattr(sage._BnReluDropoutFn, "apply", lambda *a: (calls.append(1), real(*a))[1])
    model = sage.SAGE(16, 3, 32, 3).to(dev)
    rowptr = torch.arange(0, 9, dtype=torch.int32)
    adj0 = sage.SampledAdj(rowptr.to(dev), torch.arange(8, dtype=torch.int32, device=dev), 8)
    adj1 = sage.SampledAdj(rowptr[:5].to(dev), torch.arange(4, dtype=torch.int32, device=dev), 8)
    out = model(torch.randn(8, 16, device=dev), [adj0, adj1])
    out.sum().backward()
    assert calls == [1] and out.shape == (4, 32)


@pytest.mark.parametrize("n,c", [(1550, 256), (1550, 7), (33, 3), (5, 1000)])
def test_cross_entropy_matches_torch(n, c, dev):
    """graphpope_amd.sage.cross_entropy against F.cross_entropy (main.py:216): loss and gradient, 1e-6 relative, including
    ignored labels (-100) and a non-unit upstream gradient."""
    from graphpope_amd.sage import cross_entropy
    torch.manual_seed(n + c)
    logits = (torch.randn(n, c, device=dev) * 3)
    y = torch.randint(0, c, (n,), device=dev)
    y[::7] = -100
    a, b = logits.clone().requires_grad_(True), logits.clone().requires_grad_(True)
    la, lb = F.cross_entropy(a, y), cross_entropy(b, y)
    assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(la)))
    (la * 1.7).backward(); (lb * 1.7).backward()
    _close(b.grad, a.grad, 1e-5)
    assert float(b.grad[::7].abs().max()) == 0.0                          # ignored rows get no gradient


@pytest.mark.parametrize("n,c", [(1550, 256), (1550, 7), (33, 3), (5, 1000), (4097, 64)])
def test_fused_cross_entropy_with_unit_upstream_matches_torch(n, c, dev):
    """unit_upstream=True: ONE forward launch (every block counts the labels, the last block to arrive folds the row losses)
    that already holds d(mean loss)/d(logits); loss.backward() launches nothing.  Repeated: the arrival counters reset."""
    from graphpope_amd.sage import cross_entropy
    torch.manual_seed(n * 3 + c)
    for rep in range(3):
        logits = (torch.randn(n, c, device=dev) * 3)
        y = torch.randint(0, c, (n,), device=dev)
        y[::5] = -100
        a, b = logits.clone().requires_grad_(True), logits.clone().requires_grad_(True)
        la, lb = F.cross_entropy(a, y), cross_entropy(b, y, unit_upstream=True)
        assert abs(float(la) - float(lb)) <= 1e-6 * max(1.0, abs(float(la)))
        la.backward(); lb.backward()
        _close(b.grad, a.grad, 1e-5)
        assert float(b.grad[::5].abs().max()) == 0.0
    assert torch.isnan(cross_entropy(torch.randn(8, 5, device=dev), torch.full((8,), -100, device=dev), unit_upstream=True))


def test_cross_entropy_all_ignored_and_bad_labels(dev):
    from graphpope_amd.sage import bad_label_flag, cross_entropy
    logits = torch.randn(8, 5, device=dev)
    assert torch.isnan(cross_entropy(logits, torch.full((8,), -100, device=dev)))      # torch gives nan too
    assert int(bad_label_flag(dev)) == 0
    y = torch.tensor([0, 1, 2, 3, 4, 5, 1, 1], device=dev)                # 5 is out of range: left out and flagged
    got = cross_entropy(logits, y)
    keep = torch.tensor([0, 1, 2, 3, 4, 6, 7], device=dev)
    want = F.cross_entropy(logits[keep], y[keep])
    assert abs(float(got) - float(want)) < 1e-6 and int(bad_label_flag(dev)) == 1
    bad_label_flag(dev).zero_()
