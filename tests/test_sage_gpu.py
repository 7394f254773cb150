"""HIP SAGEConv forward/backward against the torch fp32 restatement (oracle.sage_conv_torch).

PyG / torch_sparse are absent, so this parity is op-level and self-referential:
tolerances fwd 1e-4, grads 1e-3 relative to the largest magnitude (fp32 reduction-order noise).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from graphpope_amd import engine
    return engine.require_gpu()


def _random_block(n_dst, n_src, max_deg, seed, empty_rows=True):
    rs = np.random.RandomState(seed)
    deg = rs.randint(0 if empty_rows else 1, max_deg + 1, size=n_dst)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = rs.randint(0, n_src, size=int(rowptr[-1])).astype(np.int32)
    return torch.tensor(rowptr), torch.tensor(col)


def _close(got, want, rel):
    scale = max(float(want.abs().max()), 1e-6)
    err = float((got - want).abs().max())
    assert err <= rel * scale, (err, scale)


@pytest.mark.parametrize("n_dst,n_src,c_in,c_out,max_deg", [
    (5, 9, 7, 3, 3), (130, 400, 64, 32, 10), (1000, 3000, 532, 256, 25), (1550, 10136, 256, 256, 25), (300, 301, 130, 257, 4)])
def test_forward_backward_match_torch(n_dst, n_src, c_in, c_out, max_deg, dev, oracle):
    from graphpope_amd.sage import SAGEConv, SampledAdj
    rowptr, col = _random_block(n_dst, n_src, max_deg, seed=n_dst + c_in)
    torch.manual_seed(0)
    conv = SAGEConv(c_in, c_out).to(dev)
    x = torch.randn(n_src, c_in)
    g = torch.randn(n_dst, c_out)

    xd = x.to(dev).requires_grad_(True)
    out = conv((xd, xd[:n_dst]), SampledAdj(rowptr, col, n_src).to(dev))
    out.backward(g.to(dev))

    xr = x.clone().requires_grad_(True)
    wl, bl, wr = (p.detach().cpu().clone().requires_grad_(True) for p in (conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight))
    ref = oracle.sage_conv_torch(xr, rowptr, col, wl, bl, wr)
    ref.backward(g)

    _close(out.detach().cpu(), ref.detach(), 1e-4)
    _close(xd.grad.cpu(), xr.grad, 1e-3)
    _close(conv.lin_l.weight.grad.cpu(), wl.grad, 1e-3)
    _close(conv.lin_l.bias.grad.cpu(), bl.grad, 1e-3)
    _close(conv.lin_r.weight.grad.cpu(), wr.grad, 1e-3)


def test_no_input_grad_skips_grad_x(dev):
    from graphpope_amd.sage import SAGEConv, SampledAdj
    rowptr, col = _random_block(64, 200, 10, seed=1)
    conv = SAGEConv(36, 16).to(dev)
    x = torch.randn(200, 36, device=dev)                     # layer 0: features need no gradient
    out = conv((x, x[:64]), SampledAdj(rowptr, col, 200).to(dev))
    out.sum().backward()
    assert x.grad is None and conv.lin_l.weight.grad is not None


def test_state_dict_names_follow_pyg(dev):
    from graphpope_amd.sage import SAGE
    m = SAGE(756, 7, 256, 3)
    keys = set(m.state_dict())
    assert {"convs.0.lin_l.weight", "convs.0.lin_l.bias", "convs.0.lin_r.weight", "convs.2.lin_l.weight",
            "bns.0.weight", "bns.1.running_mean"} <= keys
    assert "convs.0.lin_r.bias" not in keys
    assert m.convs[0].lin_l.weight.shape == (256, 756) and m.convs[2].lin_l.weight.shape == (7, 256)


def test_model_depth_quirk_and_training_step(dev):
    """main.py:204-211 iterates over the 2 sampled adjs: with num_layers=3 the logits are hidden-wide (256)."""
    from graphpope_amd import synth
    from graphpope_amd.sage import SAGE, sample_batch
    ei, n = synth.pubmed_like()
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei[0], minlength=n))])
    n_id, adjs = sample_batch(rowptr, ei[1], np.arange(0, 512), sizes=(25, 10))
    assert adjs[1].size(0) == 512 and adjs[0].size(0) == adjs[1].size(1) and adjs[0].size(1) == len(n_id)
    torch.manual_seed(0)
    model = SAGE(40, 3, 64, 3).to(dev)
    x = torch.randn(len(n_id), 40, device=dev)
    y = torch.randint(0, 3, (512,), device=dev)
    adjs = [a.to(dev) for a in adjs]
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        out = model(x, adjs)
        assert out.shape == (512, 64)                        # hidden width, not num_classes
        loss = torch.nn.functional.cross_entropy(out, y)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.7 * losses[0]
    assert model.convs[2].lin_l.weight.grad is None          # never executed


def test_indexed_features_equal_the_materialised_batch(dev):
    """IndexedFeatures(feats, n_id) (neighbours read straight from the resident feature matrix, main.py:118-123 without the
    copy) gives bit-identical logits to x = feats[n_id] followed by the same model, and the same parameter gradients up to
    the order of the float atomics in k_scatter_mean (layer 1's grad_x, which feeds every layer-0 gradient)."""
    from graphpope_amd.sage import SAGE, IndexedFeatures, SampledAdj
    torch.manual_seed(3)
    n_all, c = 5000, 36
    feats = torch.randn(n_all, c, device=dev)
    n_id = torch.randperm(n_all, device=dev)[:900]
    rp0, col0 = _random_block(300, 900, 9, seed=1)
    rp1, col1 = _random_block(64, 300, 5, seed=2)
    adjs = [SampledAdj(rp0, col0, 900).to(dev), SampledAdj(rp1, col1, 300).to(dev)]
    a, b = SAGE(c, 4, 32, 3, dropout=0.0).to(dev), SAGE(c, 4, 32, 3, dropout=0.0).to(dev)
    b.load_state_dict(a.state_dict())
    out_a = a(feats.index_select(0, n_id), adjs)
    out_b = b(IndexedFeatures(feats, n_id), adjs)
    assert torch.equal(out_a, out_b)
    g = torch.randn_like(out_a)
    out_a.backward(g); out_b.backward(g)
    for (name, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if pa.grad is None:
            assert pb.grad is None
        else:      # 1e-5 of the gradient's scale, with a floor of 1: layer 0's bias gradient is analytically 0 under BatchNorm
            assert float((pa.grad - pb.grad).abs().max()) <= 1e-5 * max(float(pa.grad.abs().max()), 1.0), name


@pytest.fixture(scope="module")
def config2_block():
    """The outer block of a BASELINE configs[1] mini-batch: 1 550 seeds, fan-outs [25, 10] on the Flickr-shaped graph ->
    about 9 988 destinations x 37 799 sources, nnz ~ 77 k, 756 -> 256 (main.py:44, 100-116)."""
    from graphpope_amd import synth
    from graphpope_amd.sage import sample_batch
    ei, n = synth.flickr_like()
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei[0], minlength=n))])
    rng = np.random.default_rng(0)
    seeds = rng.choice(n, 1550, replace=False)
    n_id, adjs = sample_batch(rowptr, ei[1], seeds, sizes=(25, 10), rng=rng)
    a0 = adjs[0]
    assert 8000 < a0.n_dst < 12000 and 30000 < a0.n_src < 45000 and 60000 < a0.col.numel() < 110000
    return n, torch.as_tensor(n_id), a0


@pytest.fixture(params=[1, 0], ids=["gather_beside_projection", "gather_then_projection"])
def forward_order(request):
    """POPE_KNOB_SAGE_FORWARD_OVERLAP: 1 = the gather in the blocks beside the x_dst half of the projection + a second launch
    for the agg half; 0 = gather, then the whole projection."""
    from graphpope_amd import _lib
    lib = _lib.load()
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, request.param)
    yield request.param
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)


@pytest.mark.parametrize("indexed", [False, True], ids=["materialised", "indexed"])
def test_weight_gradients_with_the_xcd_aware_deal(indexed, config2_block, dev, oracle):
    """POPE_KNOB_STREAMK_XCD = 1 (the default): the stream-K weight-gradient kernel deals (group, stage) units to quads of blocks that sit in one XCD
    (csrc/gemm_streamk_tn.h); the fix-up finds each tile's partial slabs through the same mapping.  Both weight gradients against
    oracle.sage_conv_torch at bench.py's layer-0 shape and against the plain deal (the partial sums are cut at other depths: close, not
    equal), twice (deterministic), with the destination rows as a matrix and read through n_id."""
    from graphpope_amd import _lib
    from graphpope_amd.sage import SAGEConv, IndexedFeatures
    lib = _lib.load()
    n, n_id, a0 = config2_block
    c_in, c_out = 756, 256
    torch.manual_seed(0)
    feats = torch.rand(n, c_in)
    conv = SAGEConv(c_in, c_out).to(dev)
    g = torch.randn(a0.n_dst, c_out)
    adj = a0.to(dev)
    fd, nd, gd = feats.to(dev), n_id.to(dev), g.to(dev)
    grads = {}
    try:
        for order in (0, 1, 1):
            lib.pope_debug_set(_lib.KNOB_STREAMK_XCD, order)
            for p in conv.parameters():
                p.grad = None
            if indexed:
                out = conv(IndexedFeatures(fd, nd), adj)
            else:
                x = fd[nd]
                out = conv((x, x[:a0.n_dst]), adj)
            out.backward(gd)
            got = (conv.lin_l.weight.grad.clone(), conv.lin_r.weight.grad.clone())
            if order in grads:
                assert torch.equal(got[0], grads[order][0]) and torch.equal(got[1], grads[order][1])
            grads[order] = got
    finally:
        lib.pope_debug_set(_lib.KNOB_STREAMK_XCD, 1)
    xr = feats[n_id]
    wl, bl, wr = (p.detach().cpu().clone().requires_grad_(True) for p in (conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight))
    oracle.sage_conv_torch(xr, a0.rowptr, a0.col, wl, bl, wr).backward(g)
    for order in (0, 1):
        _close(grads[order][0].cpu(), wl.grad, 1e-3)
        _close(grads[order][1].cpu(), wr.grad, 1e-3)
    _close(grads[1][0].cpu(), grads[0][0].cpu(), 1e-5)
    _close(grads[1][1].cpu(), grads[0][1].cpu(), 1e-5)


@pytest.mark.parametrize("indexed", [False, True], ids=["materialised", "indexed"])
def test_config2_block_shape_matches_torch(indexed, forward_order, config2_block, dev, oracle):
    """Layer 0 at the shape bench.py times (9 988 x 37 799, 756 -> 256): forward and every gradient against
    oracle.sage_conv_torch, for the plain call and for IndexedFeatures (neighbours read through n_id, no x[n_id] copy),
    in both forward orders."""
    from graphpope_amd.sage import SAGEConv, IndexedFeatures
    n, n_id, a0 = config2_block
    c_in, c_out = 756, 256
    torch.manual_seed(0)
    feats = torch.rand(n, c_in)                                     # features (+) POPE columns are in [0, 1]
    conv = SAGEConv(c_in, c_out).to(dev)
    g = torch.randn(a0.n_dst, c_out)
    adj = a0.to(dev)
    if indexed:
        out = conv(IndexedFeatures(feats.to(dev), n_id.to(dev)), adj)
    else:
        x = feats[n_id].to(dev)
        out = conv((x, x[:a0.n_dst]), adj)
    out.backward(g.to(dev))

    xr = feats[n_id]
    wl, bl, wr = (p.detach().cpu().clone().requires_grad_(True) for p in (conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight))
    ref = oracle.sage_conv_torch(xr, a0.rowptr, a0.col, wl, bl, wr)
    ref.backward(g)
    _close(out.detach().cpu(), ref.detach(), 1e-4)
    _close(conv.lin_l.weight.grad.cpu(), wl.grad, 1e-3)
    _close(conv.lin_l.bias.grad.cpu(), bl.grad, 1e-3)
    _close(conv.lin_r.weight.grad.cpu(), wr.grad, 1e-3)


def test_config2_block_input_gradient(config2_block, dev, oracle):
    """grad_x at the config-2 block shape (hidden layers need it; layer 0 skips it): GEMM twin + scatter-mean atomics."""
    from graphpope_amd.sage import SAGEConv
    n, n_id, a0 = config2_block
    c_in, c_out = 256, 256
    torch.manual_seed(1)
    x = torch.randn(a0.n_src, c_in)
    conv = SAGEConv(c_in, c_out).to(dev)
    g = torch.randn(a0.n_dst, c_out)
    xd = x.to(dev).requires_grad_(True)
    out = conv((xd, xd[:a0.n_dst]), a0.to(dev))
    out.backward(g.to(dev))
    xr = x.clone().requires_grad_(True)
    wl, bl, wr = (p.detach().cpu().clone().requires_grad_(True) for p in (conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight))
    ref = oracle.sage_conv_torch(xr, a0.rowptr, a0.col, wl, bl, wr)
    ref.backward(g)
    _close(out.detach().cpu(), ref.detach(), 1e-4)
    _close(xd.grad.cpu(), xr.grad, 1e-3)


@pytest.mark.parametrize("n_dst,c_in,c_out", [(5800, 256, 256), (8100, 200, 256), (9988, 756, 256), (12200, 132, 256), (14300, 128, 200),
                                              (16300, 96, 128), (9988, 756, 250)])
def test_forward_projection_as_whole_tiles_fitted_to_the_chip(n_dst, c_in, c_out, dev, oracle):
    """Host-sized layer-0 shapes take gemm_tile16.h (tiles of 16 RB x 128 whose count fits the CU count: RB = 3 .. 8 over
    these row counts; ragged last row tile, a column tile cut at 250 or 200, depth padding at 132): forward against the
    torch restatement, and bit-identical to the stream-K path (POPE_KNOB_GEMM_TILE = 7) only up to summation order -- so
    against the oracle at the same 1e-4."""
    from graphpope_amd import _lib
    from graphpope_amd.sage import SAGEConv, SampledAdj
    lib = _lib.load()
    n_src = n_dst + 50
    rowptr, col = _random_block(n_dst, n_src, 4, seed=n_dst)
    torch.manual_seed(1)
    conv = SAGEConv(c_in, c_out).to(dev)
    x = torch.randn(n_src, c_in)
    adj = SampledAdj(rowptr, col, n_src).to(dev)
    with torch.no_grad():
        out = conv((x.to(dev), None), adj)
        lib.pope_debug_set(_lib.KNOB_GEMM_TILE, 7)
        try:
            out_sk = conv((x.to(dev), None), adj)
        finally:
            lib.pope_debug_set(_lib.KNOB_GEMM_TILE, 0)
        ref = oracle.sage_conv_torch(x, rowptr, col, conv.lin_l.weight.cpu(), conv.lin_l.bias.cpu(), conv.lin_r.weight.cpu())
    _close(out.cpu(), ref, 1e-4)
    _close(out_sk.cpu(), ref, 1e-4)


@pytest.mark.parametrize("n_dst,c_in,c_out,fan", [(9988, 756, 256, 8), (8100, 200, 256, 3), (10200, 132, 250, 70), (12200, 132, 256, 5)])
def test_gather_beside_the_projection_gives_the_same_aggregate_and_the_same_layer(n_dst, c_in, c_out, fan, dev):
    """The overlapped forward (launch 1: pipelined gather role beside x_dst W_r^T + b, launch 2: += agg W_l^T) against the
    sequential one: the aggregate and the copied destination rows bit for bit (same order of additions; rows of more than
    64 neighbours take the role's long-row path at fan = 70), the layer output to 1e-5 of its largest value (the two halves
    are added in another order)."""
    import ctypes
    from graphpope_amd import _lib
    lib = _lib.load()
    n_src, n_rows = n_dst + 500, n_dst + 4000
    rowptr, col = _random_block(n_dst, n_src, fan, seed=n_dst)
    g = torch.Generator().manual_seed(5)
    feats = torch.rand(n_rows, c_in, generator=g).to(dev)
    n_id = torch.randperm(n_rows, generator=g)[:n_src].to(dev)
    w_l, w_r = (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev), (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev)
    b = torch.randn(c_out, generator=g).to(dev)
    rp, cl = rowptr.to(dev), col.to(dev)
    scratch = torch.empty(max(lib.sage_conv_forward_scratch_bytes(n_dst, c_in, c_out), 16), dtype=torch.uint8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    for order in (1, 0):
        agg = torch.full((n_dst, c_in), -7.0, device=dev)
        x_dst = torch.full((n_dst, c_in), -7.0, device=dev)
        out = torch.full((n_dst, c_out), -7.0, device=dev)
        lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, order % 10)
        try:
            _lib.check(lib.sage_conv_forward_indexed(_lib.ptr(rp), _lib.ptr(cl), _lib.ptr(n_id), n_src, n_dst, cl.numel(), _lib.ptr(feats), n_rows,
                                                     c_in, _lib.ptr(w_l), _lib.ptr(b), _lib.ptr(w_r), c_out, _lib.ptr(agg), _lib.ptr(x_dst),
                                                     _lib.ptr(out), _lib.ptr(scratch), scratch.numel(), None, stream))
        finally:
            lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
        res[order] = (agg, x_dst, out)
    assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1])
    assert torch.equal(res[0][1], feats[n_id[:n_dst]])
    scale = float(res[0][2].abs().max())
    assert float((res[1][2] - res[0][2]).abs().max()) <= 1e-5 * scale
    # device extents (the captured training step): the same buffers as capacities, the true row count in a device word -- rows past
    # it are neither gathered nor waited for; the rows in front of it carry the bits of a host-sized call on that many rows
    n_true = n_dst - 137
    dims = torch.tensor([n_true, n_src, int(rowptr[n_true]), 0], dtype=torch.int32, device=dev)
    for order in (1, 0):
        got, want = [], []
        for extent in (True, False):
            agg = torch.full((n_dst, c_in), -7.0, device=dev)
            x_dst = torch.full((n_dst, c_in), -7.0, device=dev)
            out = torch.full((n_dst, c_out), -7.0, device=dev)
            lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, order)
            try:
                _lib.check(lib.sage_conv_forward_indexed(_lib.ptr(rp), _lib.ptr(cl), _lib.ptr(n_id), n_src, n_dst if extent else n_true,
                                                         cl.numel() if extent else int(rowptr[n_true]), _lib.ptr(feats), n_rows, c_in, _lib.ptr(w_l),
                                                         _lib.ptr(b), _lib.ptr(w_r), c_out, _lib.ptr(agg), _lib.ptr(x_dst), _lib.ptr(out),
                                                         _lib.ptr(scratch), scratch.numel(), _lib.ptr(dims) if extent else None, stream))
            finally:
                lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
            (got if extent else want).extend([agg, x_dst, out])
        for g_, w_ in zip(got, want):
            assert torch.equal(g_[:n_true], w_[:n_true])
            assert bool((g_[n_true:] == -7.0).all())                     # nothing written past the true extent


@pytest.mark.parametrize("n_dst,c_in,c_out", [(9988, 756, 256), (8100, 200, 256), (12200, 132, 250), (700, 40, 24)])
def test_four_stage_buffers_give_the_bits_of_three(n_dst, c_in, c_out, dev):
    """POPE_KNOB_GEMM_TILE16_BUFFERS: the whole-tile forward GEMM with four LDS stage buffers (the default: a request has two stage
    times to land) against three -- the same products added in the same order, so the layer output is the same bit for bit; the
    short last stage at depth 132 / 200 / 40 and the ragged last tiles are in the shapes."""
    from graphpope_amd import _lib
    from graphpope_amd.sage import SAGEConv, SampledAdj
    lib = _lib.load()
    n_src = n_dst + 50
    rowptr, col = _random_block(n_dst, n_src, 4, seed=n_dst)
    torch.manual_seed(3)
    conv = SAGEConv(c_in, c_out).to(dev)
    x = torch.randn(n_src, c_in).to(dev)
    adj = SampledAdj(rowptr, col, n_src).to(dev)
    outs = {}
    with torch.no_grad():
        for order in (0, 1):
            lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, order)
            try:
                for bufs in (3, 4):
                    lib.pope_debug_set(_lib.KNOB_GEMM_TILE16_BUFFERS, bufs)
                    outs[order, bufs] = conv((x, None), adj).clone()
            finally:
                lib.pope_debug_set(_lib.KNOB_GEMM_TILE16_BUFFERS, 4)
                lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
            assert torch.equal(outs[order, 3], outs[order, 4])


@pytest.mark.parametrize("n_dst,c_in,c_out", [(9988, 756, 256), (700, 40, 24)])
def test_indexed_pair_without_a_destination_matrix(n_dst, c_in, c_out, dev):
    """sage_conv_forward_indexed with x_dst = NULL and sage_conv_backward_indexed (the destination rows read through n_id by
    the projection and by the weight-gradient kernel's loader -- or, at the small shape, built in the scratch tail for the
    plain kernels): the same output and the same gradients as the pair that is given / writes the x_dst matrix."""
    import ctypes
    from graphpope_amd import _lib
    lib = _lib.load()
    n_src, n_rows = n_dst + 300, n_dst + 3000
    rowptr, col = _random_block(n_dst, n_src, 6, seed=n_dst + 1)
    g = torch.Generator().manual_seed(8)
    feats = torch.rand(n_rows, c_in, generator=g).to(dev)
    n_id = torch.randperm(n_rows, generator=g)[:n_src].to(dev)
    w_l, w_r = (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev), (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev)
    b = torch.randn(c_out, generator=g).to(dev)
    grad_out = torch.randn(n_dst, c_out, generator=g).to(dev)
    rp, cl = rowptr.to(dev), col.to(dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    nnz = cl.numel()

    def run(indexed):
        agg = torch.empty(n_dst, c_in, device=dev)
        out = torch.empty(n_dst, c_out, device=dev)
        x_dst = None if indexed else torch.empty(n_dst, c_in, device=dev)
        fbytes = lib.sage_conv_forward_indexed_scratch_bytes(n_dst, c_in, c_out)
        fs = torch.empty(max(fbytes, 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.sage_conv_forward_indexed(_lib.ptr(rp), _lib.ptr(cl), _lib.ptr(n_id), n_src, n_dst, nnz, _lib.ptr(feats), n_rows, c_in,
                                                 _lib.ptr(w_l), _lib.ptr(b), _lib.ptr(w_r), c_out, _lib.ptr(agg), _lib.ptr(x_dst), _lib.ptr(out),
                                                 _lib.ptr(fs), fbytes, None, stream))
        gwl, gwr, gb = torch.empty_like(w_l), torch.empty_like(w_r), torch.empty(c_out, device=dev)
        if indexed:
            bbytes = lib.sage_conv_backward_indexed_scratch_bytes(n_src, n_dst, nnz, c_in, c_out)
            bs = torch.empty(bbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.sage_conv_backward_indexed(_lib.ptr(rp), _lib.ptr(cl), _lib.ptr(n_id), n_src, n_dst, nnz, _lib.ptr(feats), n_rows, _lib.ptr(agg),
                                                      c_in, _lib.ptr(w_l), _lib.ptr(w_r), c_out, _lib.ptr(grad_out), _lib.ptr(gwl), _lib.ptr(gb),
                                                      _lib.ptr(gwr), _lib.ptr(bs), bbytes, None, stream))
        else:
            bbytes = lib.sage_conv_scratch_bytes(n_dst, n_dst, nnz, c_in, c_out)
            bs = torch.empty(bbytes, dtype=torch.uint8, device=dev)
            _lib.check(lib.sage_conv_backward(_lib.ptr(rp), _lib.ptr(cl), n_dst, n_dst, nnz, _lib.ptr(x_dst), _lib.ptr(agg), c_in, _lib.ptr(w_l),
                                              _lib.ptr(w_r), c_out, _lib.ptr(grad_out), None, _lib.ptr(gwl), _lib.ptr(gb), _lib.ptr(gwr),
                                              _lib.ptr(bs), bbytes, None, stream))
        torch.cuda.synchronize()
        return out, gwl, gwr, gb

    a, b_ = run(True), run(False)
    assert torch.equal(a[0], b_[0])                               # the forward pass takes the same kernels either way
    for x, y in zip(a[1:], b_[1:]):                               # stream-K deals the same units: bit for bit the same sums
        _close(x, y, 1e-6)
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 0)           # the sequential order wants the matrix: without x_dst it needs the larger scratch
    try:
        with pytest.raises(_lib.PopeError, match="scratch"):
            _lib.check(lib.sage_conv_forward_indexed(_lib.ptr(rp), _lib.ptr(cl), _lib.ptr(n_id), n_src, n_dst, nnz, _lib.ptr(feats), n_rows, c_in,
                                                     _lib.ptr(w_l), _lib.ptr(b), _lib.ptr(w_r), c_out, _lib.ptr(torch.empty(n_dst, c_in, device=dev)),
                                                     None, _lib.ptr(torch.empty(n_dst, c_out, device=dev)), None, 0, None, stream))
    finally:
        lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
