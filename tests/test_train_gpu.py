"""Device-extent batches and the replayed training step (graphpope_amd.train) against the host-sized autograd path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def graph():
    from graphpope_amd import engine, synth
    dev = engine.require_gpu()
    ei = synth.powerlaw_graph(6000, 40000, seed=7, alpha=0.9, shift=0.8)
    csr = engine.build_csr(torch.as_tensor(ei, device=dev), 6000)
    return dev, ei, csr


@pytest.mark.parametrize("sizes,seed", [((25, 10), 1), ((3,), 99), ((4, 3, 2), 2**40 + 7)])
def test_device_extent_sampler_equals_the_host_sized_one(graph, sizes, seed):
    """sage_sample_batch_device (no read-back, buffers at capacity, sizes in device words) draws exactly the batch
    sage_sample_batch draws -- which tests/test_sampler_gpu.py pins bit for bit against the CPU restatement."""
    from graphpope_amd.sampler import NeighborSampler
    dev, _, csr = graph
    seeds = torch.as_tensor(np.random.RandomState(3).choice(6000, 257, replace=False), device=dev)
    s = NeighborSampler(csr.rowptr, csr.col, 6000, sizes)
    n_id, adjs = s.sample(seeds, seed=seed)
    for split in (0, 12345):                                   # part of the seed in the device word
        word = torch.tensor([split], dtype=torch.int64, device=dev)
        b = s.sample_device(seeds, seed=seed - split, seed_dev=word)
        dims = b.dims.cpu().numpy()
        assert dims[-1, 1] == n_id.numel() and torch.equal(b.n_id[: n_id.numel()], n_id)
        for adj, want, d in zip(b.adjs, adjs, dims[::-1]):
            assert tuple(d[:3]) == (want.n_dst, want.n_src, want.col.numel())
            assert torch.equal(adj.rowptr[: want.n_dst + 1], want.rowptr) and torch.equal(adj.col[: want.col.numel()], want.col)
            assert bool((adj.rowptr[want.n_dst:] == want.col.numel()).all())        # rows past n_dst are empty
    b2 = s.sample_device(seeds, seed=seed, out=b)              # refilled in place
    assert b2 is b and torch.equal(b.n_id[: n_id.numel()], n_id)


def _model(dev, c_in=40, hidden=48, layers=3):
    from graphpope_amd.sage import SAGE
    torch.manual_seed(0)
    return SAGE(c_in, 5, hidden, layers).to(dev)


@pytest.mark.parametrize("sizes,layers,c_in,hidden", [((25, 10), 3, 40, 48), ((6, 4, 3), 4, 40, 48), ((5, 3), 3, 37, 30)],
                         ids=["two_hops", "three_hops", "odd_widths"])
def test_device_extent_layers_equal_the_host_sized_ones(graph, sizes, layers, c_in, hidden):
    """Forward, loss and every gradient of the model on a DeviceBatch (capacity-shaped tensors, sizes read on the device)
    against the same batch with host-known sizes.  Three hops: the middle layer has BOTH extents on the device (tile GEMMs
    with a device row count forward, a device depth in the weight gradients); odd widths: the scalar kernel variants.
    Dropout off: the masks are functions of the element index and the row pitch is the same, but the comparison should
    not depend on that."""
    from graphpope_amd.sage import IndexedFeatures, cross_entropy
    from graphpope_amd.sampler import NeighborSampler
    dev, _, csr = graph
    feats = torch.randn(6000, c_in, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    seeds = torch.arange(100, 612, device=dev)
    y = torch.randint(0, 5, (512,), device=dev)
    s = NeighborSampler(csr.rowptr, csr.col, 6000, sizes)
    n_id, adjs = s.sample(seeds, seed=5)
    b = s.sample_device(seeds, seed=5)
    res = []
    for x, a in ((IndexedFeatures(feats, n_id), adjs), (IndexedFeatures(feats, b.n_id), b.adjs)):
        m = _model(dev, c_in, hidden, layers)
        m.dropout = 0.0
        out = m(x, a)
        loss = cross_entropy(out, y)
        loss.backward()
        res.append((out.detach(), loss.detach(), [p.grad.clone() for p in m.parameters() if p.grad is not None],
                    [bn.running_var.clone() for bn in m.bns[:1]]))
    (o0, l0, g0, r0), (o1, l1, g1, r1) = res
    assert torch.allclose(o0, o1[: o0.shape[0]], rtol=1e-5, atol=1e-6) and torch.allclose(l0, l1, rtol=1e-6)
    assert len(g0) == len(g1) and len(g0) >= 3 * len(sizes)
    for a, c in zip(g0, g1):
        assert torch.allclose(a, c, rtol=1e-3, atol=1e-5)      # the scatter's float atomics land in a different order
    assert torch.allclose(r0[0], r1[0], rtol=1e-5)


@pytest.mark.parametrize("with_sampler", [True, False], ids=["sampler_in_graph", "presampled_pool"])
def test_replayed_step_equals_the_eager_step(graph, with_sampler):
    """SageTrainStep with graph=True (captured once, replayed) against graph=False (the same body enqueued eagerly every
    step): same seeds -> same losses and the same parameters after 8 steps, to float-atomic noise."""
    from graphpope_amd.optim import Adam
    from graphpope_amd.sampler import DeviceBatch, NeighborSampler
    from graphpope_amd.train import SageTrainStep
    dev, _, csr = graph
    feats = torch.randn(6000, 40, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    labels = torch.randint(0, 5, (6000,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    sampler = NeighborSampler(csr.rowptr, csr.col, 6000, (25, 10))
    perm = torch.randperm(6000, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    pool = []
    if not with_sampler:
        for i in range(4):
            sd = perm[i * 256:(i + 1) * 256].contiguous()
            n_id, adjs = sampler.sample(sd, seed=i)
            db = DeviceBatch(256, (25, 10), dev)
            db.load(n_id, adjs)
            pool.append((db, labels[sd].contiguous()))
    runs = []
    for use_graph in (False, True):
        m = _model(dev)
        opt = Adam(m.parameters(), lr=0.01)
        st = SageTrainStep(m, opt, feats, 256, (25, 10), sampler=sampler if with_sampler else None, clip=0.5, graph=use_graph, seed=11)
        losses = []
        for i in range(8):
            if with_sampler:
                sd = perm[(i % 6) * 256:(i % 6 + 1) * 256].contiguous()
                losses.append(st.step(sd, labels[sd].contiguous()).item())
            else:
                db, yb = pool[i % 4]
                st.load_batch(db, yb)
                losses.append(st.run().item())
        runs.append((losses, [p.detach().clone() for p in m.parameters()], int(st.state.adam_step.item()),
                     opt.state_dict()["state"][0]["step"].item(), [bn.num_batches_tracked.item() for bn in m.bns[:1]]))
    (la, pa, sa, ha, na), (lb, pb, sb, hb, nb) = runs
    assert sa == sb == 9 and ha == hb == 8.0 and na == nb == [8]
    assert np.allclose(la, lb, rtol=1e-4) and la[-1] < la[0]
    # Adam divides by sqrt(v): where a gradient is noise around zero (the bias in front of a training-mode BatchNorm has a
    # zero gradient in exact arithmetic), float-atomic reordering moves a parameter by up to lr per step, so the parameters
    # are compared as vectors with a loose bound (the losses above pin the trajectory step by step)
    for a, c in zip(pa, pb):
        assert float((a - c).norm()) <= 0.1 * float(a.norm()) + 1e-6


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "replayed"])
def test_epoch_mode_draws_the_batches_a_loader_would(graph, use_graph):
    """set_epoch / step_epoch (seeds = consecutive slices of the epoch's order read through a device cursor, labels gathered
    by the sampler's first kernel: sage_sample_epoch_batch_device) against step(seeds, labels[seeds]) fed by the host with
    the same slices: the same batches (node lists bit for bit), the same labels, the same losses -- over two epochs with
    different orders, the second one shorter."""
    from graphpope_amd.optim import Adam
    from graphpope_amd.sampler import NeighborSampler
    from graphpope_amd.train import SageTrainStep
    dev, _, csr = graph
    feats = torch.randn(6000, 40, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    labels = torch.randint(0, 5, (6000,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    sampler = NeighborSampler(csr.rowptr, csr.col, 6000, (25, 10))
    orders = [torch.randperm(6000, device=dev, generator=torch.Generator(device=dev).manual_seed(3))[:1300],
              torch.randperm(6000, device=dev, generator=torch.Generator(device=dev).manual_seed(4))[:800]]
    runs = []
    for epoch_mode in (False, True):
        m = _model(dev)
        opt = Adam(m.parameters(), lr=0.01)
        st = SageTrainStep(m, opt, feats, 256, (25, 10), sampler=sampler, clip=0.5, graph=use_graph, seed=11)
        losses, ids, ys = [], [], []
        for order in orders:
            if epoch_mode:
                st.set_epoch(order, labels)
                assert st.batches_left() == order.numel() // 256
                while st.batches_left():
                    losses.append(st.step_epoch().item())
                    ids.append(st.batch.n_id[: int(st.batch.dims[-1][1])].clone())      # the outermost block's node list
                    ys.append(st.y.clone())
                with pytest.raises(AssertionError):
                    st.step_epoch()                                  # the short tail is the caller's
            else:
                for b in range(order.numel() // 256):
                    sd = order[b * 256:(b + 1) * 256].contiguous()
                    losses.append(st.step(sd, labels[sd].contiguous()).item())
                    ids.append(st.batch.n_id[: int(st.batch.dims[-1][1])].clone())
                    ys.append(st.y.clone())
        runs.append((losses, ids, ys))
    (la, ia, ya), (lb, ib, yb) = runs
    assert len(la) == len(lb) == 5 + 3
    for a, b in zip(ia, ib):
        assert torch.equal(a, b)
    for a, b in zip(ya, yb):
        assert torch.equal(a, b)
    assert np.allclose(la, lb, rtol=1e-4)


def test_adam_device_step_matches_the_host_step(graph):
    from graphpope_amd.optim import Adam
    dev = graph[0]
    torch.manual_seed(0)
    w0 = torch.randn(1000, device=dev)
    grads = [torch.randn(1000, device=dev) for _ in range(5)]
    outs = []
    for device_word in (False, True):
        p = torch.nn.Parameter(w0.clone())
        opt = Adam([p], lr=0.01)
        word = torch.ones(1, dtype=torch.int64, device=dev)
        if device_word:
            opt.use_device_step(word)
        for g in grads:
            p.grad = g.clone()
            opt.step()
            word += 1
        outs.append(p.detach().clone())
    assert torch.allclose(outs[0], outs[1], rtol=1e-6, atol=1e-7)


def test_sampling_ahead_on_a_side_stream_gives_the_same_steps(graph):
    """prefetch=True (the next batch is sampled on a side stream while the step computes, two batch buffers) against the
    same steps with the sampler in line: same seeds, same sampling-seed sequence -> same losses."""
    from graphpope_amd.optim import Adam
    from graphpope_amd.sampler import NeighborSampler
    from graphpope_amd.train import SageTrainStep
    dev, _, csr = graph
    feats = torch.randn(6000, 40, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    labels = torch.randint(0, 5, (6000,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    sampler = NeighborSampler(csr.rowptr, csr.col, 6000, (25, 10))
    perm = torch.randperm(6000, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    batches = [perm[i * 256:(i + 1) * 256].contiguous() for i in range(10)]
    runs = []
    for prefetch in (False, True):
        m = _model(dev)
        st = SageTrainStep(m, Adam(m.parameters(), lr=0.01), feats, 256, sampler=sampler, graph=False, seed=5, prefetch=prefetch)
        losses = []
        for i, sd in enumerate(batches):
            nxt = batches[i + 1] if i + 1 < len(batches) else None
            losses.append(st.step(sd, labels[sd].contiguous(), nxt, None if nxt is None else labels[nxt].contiguous()).item())
        runs.append(losses)
    assert np.allclose(runs[0], runs[1], rtol=1e-4) and runs[0][-1] < runs[0][0]


def test_a_captured_step_keeps_its_own_sampler_scratch(graph):
    """ADVICE (round 3): the captured step has the sampler's scratch address baked in, and the host-sized sampling paths used
    to share it and replace it when they needed a larger one -- the next replay then wrote into freed memory.  The
    device-extent paths now take their scratch from the DeviceBatch (never replaced): a replayed trainer whose sampler serves
    a much larger host-sized batch (and a second, larger DeviceBatch) in between draws the same batches and the same losses
    as an undisturbed one."""
    from graphpope_amd.optim import Adam
    from graphpope_amd.sampler import DeviceBatch, NeighborSampler
    from graphpope_amd.train import SageTrainStep
    dev, _, csr = graph
    feats = torch.randn(6000, 40, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    labels = torch.randint(0, 5, (6000,), device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    perm = torch.randperm(6000, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    runs = []
    for disturb in (False, True):
        sampler = NeighborSampler(csr.rowptr, csr.col, 6000, (25, 10))
        m = _model(dev)
        opt = Adam(m.parameters(), lr=0.01)
        st = SageTrainStep(m, opt, feats, 128, (25, 10), sampler=sampler, clip=0.5, graph=True, seed=11)
        losses, ids = [], []
        for i in range(8):
            if disturb and i >= 3:
                big = perm[: 1024 + 256 * i].contiguous()             # every call needs more scratch than the one before
                sampler.sample(big, seed=i)
                sampler.sample_device(perm[: 512].contiguous(), seed=i, out=DeviceBatch(512, (25, 10), dev))
                junk = torch.full((1 << 22,), 0x7F, dtype=torch.uint8, device=dev)   # whatever was freed gets reused and overwritten
                del junk
            sd = perm[(i % 6) * 128:(i % 6 + 1) * 128].contiguous()
            losses.append(st.step(sd, labels[sd].contiguous()).item())
            ids.append(st.batch.n_id[: int(st.batch.dims[-1][1])].clone())
        runs.append((losses, ids))
    (la, ia), (lb, ib) = runs
    for a, b in zip(ia, ib):
        assert torch.equal(a, b)
    assert np.allclose(la, lb, rtol=1e-4)
