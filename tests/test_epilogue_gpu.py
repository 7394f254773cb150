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


@pytest.mark.parametrize("n_dst,c_in,c_out,extent", [(9988, 756, 256, False), (9988, 756, 256, True), (8100, 200, 256, False), (700, 40, 24, False)])
@pytest.mark.parametrize("indexed", [False, True], ids=["materialised", "indexed"])
def test_batchnorm_statistics_out_of_the_projections_epilogue(n_dst, c_in, c_out, extent, indexed, dev):
    """main.py:206-209: x = convs[i](...); x = bns[i](x); relu; dropout.  SAGEConv(..., bn_stats=True) lets the projection's epilogue
    produce the first stage of the BatchNorm statistics (float64 column sums per row tile) and bn_relu_dropout picks them up: one
    launch less.  Against the same two modules without it: the conv output bit for bit, the BatchNorm output, its running
    statistics and every gradient to 1e-6 of their scale (the float64 sums are added in another order); with a device extent
    (capacity rows behind the true count are neither produced nor counted); at a shape whose kernels produce no statistics the
    three-launch form runs (no _bn_stats on the tensor)."""
    import copy
    from graphpope_amd.sage import IndexedFeatures, SAGEConv, SampledAdj, bn_relu_dropout
    g = torch.Generator().manual_seed(n_dst + c_in)
    n_src = n_dst + 300
    deg = torch.randint(1, 9, (n_dst,), generator=g)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64), deg.cumsum(0)]).to(torch.int32)
    col = torch.randint(0, n_src, (int(rowptr[-1]),), generator=g, dtype=torch.int32)
    torch.manual_seed(1)
    conv = SAGEConv(c_in, c_out).to(dev)
    bn_a = torch.nn.BatchNorm1d(c_out).to(dev)
    bn_b = copy.deepcopy(bn_a)
    n_true = n_dst - 77 if extent else n_dst
    dims = torch.tensor([n_true, n_src, int(rowptr[n_true]), 0], dtype=torch.int32, device=dev) if extent else None
    adj = SampledAdj(rowptr.to(dev), col.to(dev), n_src, dims)
    if indexed:
        feats = torch.rand(n_src + 1000, c_in, generator=g).to(dev)
        n_id = torch.randperm(n_src + 1000, generator=g)[:n_src].to(dev)
        x = IndexedFeatures(feats, n_id)
    else:
        xs = torch.rand(n_src, c_in, generator=g).to(dev)
        x = (xs, xs[:n_dst])
    rows = None if dims is None else dims[0:1]
    res = []
    for stats, bn in ((False, bn_a), (True, bn_b)):
        conv.zero_grad(set_to_none=True)
        h = conv(x, adj, bn_stats=stats)
        produced = hasattr(h, "_bn_stats")
        y = bn_relu_dropout(h, bn, 0.5, True, seed=1234, rows=rows)
        (y[:n_true] * torch.linspace(0.5, 1.5, c_out, device=dev)).sum().backward()
        res.append((h.detach(), y.detach(), bn.running_mean.clone(), bn.running_var.clone(), bn.weight.grad.clone(), bn.bias.grad.clone(),
                    conv.lin_l.weight.grad.clone(), conv.lin_l.bias.grad.clone(), produced))
    a, b = res
    assert not a[8] and b[8] == (n_dst * c_out >= 64 * 1024)                 # tiny products stay on kernels without the epilogue
    assert torch.equal(a[0][:n_true], b[0][:n_true])
    for u, v in zip(a[1:7], b[1:7]):
        u, v = (u[:n_true], v[:n_true]) if u.dim() == 2 and u.shape[0] == n_dst else (u, v)
        assert float((u - v).abs().max()) <= 1e-6 * max(float(u.abs().max()), 1e-3)
    # the conv's bias gradient = column sums of BatchNorm's input gradient: zero in exact arithmetic (BatchNorm removes what a bias
    # adds), rounding noise in both forms -- summed from the float32 matrix (a) or taken from the float64 sums (b, round 5)
    scale = float(a[6].abs().max())
    assert float(a[7].abs().max()) <= 1e-4 * scale and float(b[7].abs().max()) <= 1e-4 * scale
    assert int(bn_a.num_batches_tracked) == int(bn_b.num_batches_tracked) == 1


@pytest.mark.parametrize("m,c,training", [(9988, 256, 1), (9988, 256, 0), (777, 36, 1), (777, 37, 0)])
def test_column_sums_of_the_input_gradient_from_the_statistics_pass(m, c, training, dev):
    """sage_bn_relu_dropout_backward_bias: grad_x_colsum[c] = sum over the rows of grad_x[:, c] (the bias gradient of the layer in front,
    main.py:206-207) out of the backward statistics' float64 sums instead of a pass over grad_x.  Against the float64 column sums of
    the grad_x matrix the same call writes: equal to 1e-6 of the sum of |grad_x| per column (in training mode the true value is zero
    and both are rounding noise; in eval mode it is gamma * rstd * sum g); grad_x, grad_gamma, grad_beta are those of the plain call."""
    import ctypes
    from graphpope_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(m + c)
    x = (torch.randn(m, c, generator=g) * 2 + 0.5).to(dev)
    dy = torch.randn(m, c, generator=g).to(dev)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(dev), torch.randn(c, generator=g).to(dev)
    mean, rstd = x.mean(0), 1.0 / torch.sqrt(x.var(0, unbiased=False) + 1e-5)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    scratch = torch.empty(lib.sage_bn_scratch_bytes(c), dtype=torch.uint8, device=dev)
    outs = []
    for with_bias in (False, True):
        gx, gg, gb = torch.empty_like(x), torch.empty(c, device=dev), torch.empty(c, device=dev)
        cs = torch.full((c,), float("nan"), device=dev)
        args = (_lib.ptr(x), _lib.ptr(dy), m, c, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(mean), _lib.ptr(rstd), training, 0.5, 99, _lib.ptr(gx),
                _lib.ptr(gg), _lib.ptr(gb), _lib.ptr(scratch), scratch.numel(), None, None)
        _lib.check(lib.sage_bn_relu_dropout_backward_bias(*args, _lib.ptr(cs), stream) if with_bias else lib.sage_bn_relu_dropout_backward(*args, stream))
        outs.append((gx, gg, gb, cs))
    assert all(torch.equal(u, v) for u, v in zip(outs[0][:3], outs[1][:3]))
    gx, cs = outs[1][0].double(), outs[1][3].double()
    assert float((cs - gx.sum(0)).abs().max()) <= 1e-6 * float(gx.abs().sum(0).max())
    if not training:
        assert float(cs.abs().max()) > 1e-3 * float(gx.abs().sum(0).max())        # a real quantity there, not noise


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


@pytest.mark.parametrize("n,c,with_params", [(1550, 256, True), (33, 3, True), (4097, 64, False)])
def test_cross_entropy_finished_by_the_optimisers_launch(n, c, with_params, dev):
    """cross_entropy(..., unit_upstream=True, loss_in=opt): the mean of the row losses is one more block of opt.step()'s launch
    (sage_adam_step_loss) instead of a launch in front of the backward pass.  After the step: the loss F.cross_entropy gives
    (main.py:216), the gradient of the mean loss, and bit for bit the parameters the plain one-launch step (sage_adam_step, itself
    checked against torch.optim.Adam in test_optim_gpu.py; main.py:244) makes of the same gradients -- also when no parameter has a
    gradient (the loss is then finished on its own)."""
    from graphpope_amd.optim import Adam
    from graphpope_amd.sage import cross_entropy
    torch.manual_seed(n + 5 * c)
    w = (torch.randn(c, c, device=dev) * 0.1)
    mine, plain, ref = torch.nn.Parameter(w.clone()), torch.nn.Parameter(w.clone()), torch.nn.Parameter(w.clone())
    opt, popt = Adam([mine], lr=1e-2), Adam([plain], lr=1e-2)
    for rep in range(3):
        x = torch.randn(n, c, device=dev) * 2
        y = torch.randint(0, c, (n,), device=dev)
        y[::6] = -100
        opt.zero_grad(set_to_none=True)
        if with_params:
            ref.data.copy_(mine.data)
            ref.grad = None
            lm = cross_entropy(x @ mine, y, unit_upstream=True, loss_in=opt)
            lm.backward()
            lr_ = F.cross_entropy(x @ ref, y)
            lr_.backward()
            _close(mine.grad, ref.grad, 1e-4)
            plain.grad = mine.grad.clone()
            opt.step(); popt.step()
            assert torch.equal(mine.detach(), plain.detach())
        else:
            lm = cross_entropy(x, y, unit_upstream=True, loss_in=opt)
            lr_ = F.cross_entropy(x, y)
            opt.step()                                                        # nothing to update: the loss is finished all the same
        assert abs(float(lm) - float(lr_)) <= 1e-6 * max(1.0, abs(float(lr_))), rep
    with pytest.raises(ValueError):
        cross_entropy(torch.randn(4, 3, device=dev), torch.zeros(4, dtype=torch.int64, device=dev), loss_in=opt)


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
