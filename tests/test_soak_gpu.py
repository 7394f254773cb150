"""Randomised soaks (GPU): the level kernel's instantiations against the oracle on small random graphs, every finalise kernel against
the generic one on random shapes, the host -> host result assembly through every transport, Graphpope() calls in a row, the fan-out
sampler against its CPU restatement, the node2vec embedding against the oracle.  GRAPHPOPE_SOAK_SEEDS scales all of them.

The large-graph forms of k_bfs_level (live table read from global memory, with or without its LDS summary; 8-word tiles; tiles walked
inside the wave, in pairs) only run by themselves on graphs of more than 256 Ki nodes, where the oracle needs seconds per case.
POPE_KNOB_LIVE_MODE forces them on small graphs: random directed multigraphs with self-loops, isolated nodes, long paths (deep levels)
and hubs (rows spanning many 256-slot chunks), anchors drawn with repeats, 1 to 1 100 anchors (1 to 18 words per node), on workspaces
that hold the previous case's bytes.  utils.py:64-81 (nx.shortest_path per node and anchor) is what the hop counts must equal.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEEDS = int(os.environ.get("GRAPHPOPE_SOAK_SEEDS", "6"))
BASE = int(os.environ.get("GRAPHPOPE_SOAK_BASE", "0"))         # first seed (a later soak continues where an earlier one stopped)        # a long soak: GRAPHPOPE_SOAK_SEEDS=200 pytest tests/test_soak_gpu.py


def _graph(rs, kind):
    if kind == "sparse":                         # many components, unreachable pairs, isolated nodes
        n = int(rs.randint(50, 3000))
        e = int(n * rs.uniform(0.5, 2.0))
        ei = rs.randint(0, n, (2, e))
    elif kind == "hubs":                         # a few rows of thousands of slots
        n = int(rs.randint(600, 4000))
        hubs = rs.randint(0, n, 3)
        src = np.concatenate([np.repeat(hubs, 1500), rs.randint(0, n, 4 * n)])
        dst = np.concatenate([rs.randint(0, n, 4500), rs.randint(0, n, 4 * n)])
        ei = np.stack([src, dst])
        ei = np.concatenate([ei, ei[::-1]], 1)
    elif kind == "path":                         # depth beyond the four eagerly cleared hop-bit planes
        n = int(rs.randint(40, 400))
        a = np.arange(n - 1)
        ei = np.stack([np.concatenate([a, a + 1]), np.concatenate([a + 1, a])])
        extra = rs.randint(0, n, (2, n // 10))
        ei = np.concatenate([ei, extra], 1)
    else:                                        # dense-ish, with self-loops and repeated edges
        n = int(rs.randint(200, 2500))
        ei = rs.randint(0, n, (2, 12 * n))
        ei[:, : n // 20] = ei[0, : n // 20]
    if rs.rand() < 0.5:
        ei = ei[:, np.lexsort((ei[1], ei[0]))]   # the sorted fast path of the CSR build; else the counting path
    return ei.astype(np.int64), n


@pytest.mark.parametrize("seed", range(SEEDS))
def test_forced_live_modes_and_tilings_on_random_graphs(seed, oracle):
    from graphpope_amd import _lib, engine
    dev = engine.require_gpu()
    lib = _lib.load()
    rs = np.random.RandomState(1000 + BASE + seed)
    ks = [1, 64, 65, 128, 200, 256, 257, 300, 511, 512, 513, 600, 768, 1000, 1024, 1100]
    try:
        for case in range(10):
            ei_np, n = _graph(rs, ["sparse", "hubs", "path", "dense"][(case + seed) % 4])
            k = int(ks[rs.randint(len(ks))])
            anchors = rs.randint(0, n, k)                                 # with repeats
            want = oracle.geodesic_hops(ei_np, n, anchors)
            eid = torch.as_tensor(ei_np, device=dev)
            for mode in (-1, 2, 3):
                _lib.check(lib.pope_debug_set(_lib.KNOB_LIVE_MODE, mode))
                _, hp = engine.geodesic_run(None, eid, n, anchors, want_out=False, reuse_workspace=True)
                got = engine.hop_matrix(hp).cpu().numpy()
                assert np.array_equal(got, want), (seed, case, n, ei_np.shape[1], k, mode)
    finally:
        lib.pope_debug_set(_lib.KNOB_LIVE_MODE, -1)


@pytest.mark.parametrize("seed", range(max(4, SEEDS // 2)))
def test_finalise_kernels_agree_on_random_shapes(seed):
    """Every finalise kernel the library can pick (pipelined, shuffle, LDS-table in both batch orders, rounds 1-3) against the generic one
    (POPE_KNOB_FINALIZE_VARIANT 0) on random planes of random shapes: node counts that leave ragged last batches, feature widths with
    and without 16-byte rows, 1-8 shards of 4 to 2 048 anchors, 0 to 4 hop bits, a column offset and spare columns behind the
    embedding; NaN-poisoned outputs, so a missed or a doubly-claimed element shows (utils.py:73, 129-135: out = x (+) 1 / (hops + 1))."""
    from graphpope_amd import _lib, engine
    dev = engine.require_gpu()
    lib = _lib.load()
    rs = np.random.RandomState(7000 + BASE + seed)
    g = torch.Generator().manual_seed(BASE + seed)
    try:
        for case in range(14):
            n = int(rs.choice([1, 5, 63, 64, 65, 257, 1000, 4099, 12345]))
            shards = int(rs.choice([1, 1, 2, 3, 4, 8]))
            k = int(rs.choice([4, 7, 60, 64, 100, 128, 192, 256, 320, 512, 1024, 2048]))
            if shards > 1 and k * shards > 4096:
                k = 256
            f = int(rs.choice([0, 0, 3, 16, 100, 500, 1028]))
            bits = int(rs.choice([0, 1, 3, 4]))
            spare = int(rs.choice([0, 0, 4, 5])) if shards == 1 else 0
            c0 = int(rs.choice([0, 0, 4, 64])) if shards == 1 else 0
            w = lib.pope_words(k)
            planes = torch.randint(-2**62, 2**62, (shards, 5, n, w), generator=g, dtype=torch.int64).to(dev)
            x = torch.rand((n, f), generator=g).to(dev) if f else None
            outs = []
            for variant in (0, 8, 9, 10, 12, 7):
                lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 9 if variant == 12 else variant)
                lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 12 if variant == 12 else 11)
                out = torch.full((n, f + c0 + shards * k + spare), float("nan"), device=dev)
                if shards == 1:
                    engine.finalize(planes[0], bits, n, k, x, f, out, c0)
                else:
                    engine.finalize_shards(planes, bits, n, k, x, f, out)
                outs.append(out)
            ref = outs[0]
            written = torch.ones_like(ref, dtype=torch.bool)
            written[:, f:f + c0] = False
            if spare:
                written[:, -spare:] = False
            assert not torch.isnan(ref[written]).any() and torch.isnan(ref[~written]).all(), (seed, case)
            for o in outs[1:]:
                assert torch.equal(torch.nan_to_num(o, nan=-7.0), torch.nan_to_num(ref, nan=-7.0)), (seed, case, n, shards, k, f, bits, c0, spare)
    finally:
        lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 8)
        lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 11)


@pytest.mark.parametrize("seed", range(max(3, SEEDS // 4)))
def test_host_results_assemble_on_random_shapes(seed):
    """The host -> host boundary's last step (utils.py:129-135 on the host: features (+) embedding into a pageable tensor) on random
    shapes through every transport round 5 left -- pinned ring, bounce buffer (ring refused), blocking copies (both refused) -- as
    floats and as byte codes, into heap tensors and page-aligned mappings, with strided sources, 1-7 copy threads and 0-9 chunks; the
    result is poisoned first and must equal the sources bit for bit."""
    from graphpope_amd import _lib, engine
    dev = engine.require_gpu()
    lib = _lib.load()
    rs = np.random.RandomState(9000 + BASE + seed)
    g = torch.Generator().manual_seed(BASE + seed)
    lut = torch.rand(256, generator=g).to(dev)
    try:
        for case in range(8):
            n = int(rs.choice([1, 3, 17, 1000, 4097, 30011, 120001]))
            f = int(rs.choice([0, 1, 3, 40, 500]))
            k = int(rs.choice([1, 5, 36, 64, 255, 256, 1000]))
            refuse = int(rs.choice([0, 0, 2, 6]))
            threads, chunks = int(rs.randint(1, 8)), int(rs.randint(0, 10))
            coded = bool(rs.randint(2)) and refuse == 0           # the byte transport needs the pinned ring (utils.py falls back to floats without it)
            lib.pope_debug_set(_lib.KNOB_FAIL_HOST_REGISTER, refuse)
            pad = int(rs.choice([0, 3]))
            x = torch.rand(n, f + pad, generator=g)[:, :f] if f else None
            out = engine.host_result_tensor(n, f + k) if rs.randint(2) else torch.empty((n, f + k))
            out.fill_(-1.0)
            if coded:
                codes = torch.randint(0, 256, (n, k + pad), generator=g, dtype=torch.uint8).to(dev)[:, :k]
                with engine.HostAssembly(x, out, f, threads=threads) as asm:
                    asm.finish_codes(codes, lut)
                want = lut.cpu()[codes.cpu().long()]
            else:
                emb = torch.rand(n, k + pad, generator=g).to(dev)[:, :k]
                engine.assemble_host_result(x, emb, out, f, threads=threads, chunks=chunks)
                want = emb.cpu()
            assert (f == 0 or torch.equal(out[:, :f], x)) and torch.equal(out[:, f:], want), (seed, case, n, f, k, refuse, threads, chunks, coded)
            del out
    finally:
        lib.pope_debug_set(_lib.KNOB_FAIL_HOST_REGISTER, 0)


def test_graphpope_calls_in_a_row_leak_nothing_and_stay_exact(oracle):
    """The host -> host boundary as a process uses it (utils.py:182-210): Graphpope() calls over three graph sizes and both
    transports (ring / staged), results released and re-allocated in between, ordinary pageable torch transfers of recycled host
    buffers interleaved (the kind of transfer that died behind an assembly in rounds 3-4) -- every eighth result bit for bit
    against the oracle; resident memory, device memory and open files must not grow.  24 calls by default, GRAPHPOPE_SOAK_SEEDS x 4."""
    import contextlib
    import gc
    import sys
    import psutil
    from graphpope_amd import engine, synth, utils as gp
    dev = engine.require_gpu()
    proc = psutil.Process()
    cases = []
    for scale, f, k in ((12, 20, 96), (14, 64, 128), (15, 100, 256)):
        ei, n = synth.rmat(scale, edge_factor=8, seed=scale)
        cases.append((ei, n, np.random.RandomState(scale).rand(n, f).astype(np.float32), k))

    class Data:
        pass

    calls = max(24, 4 * SEEDS)
    marks = []
    old = os.environ.get("GRAPHPOPE_HOST_RESULT")
    try:
        for it in range(calls):
            ei, n, x, k = cases[it % 3]
            os.environ["GRAPHPOPE_HOST_RESULT"] = ("ring", "staged")[(it // 3) % 2]
            d = Data()
            d.x, d.edge_index, d.num_nodes = torch.as_tensor(x), torch.as_tensor(ei), n
            gp.clear_cache()
            np.random.seed(BASE + it)
            with contextlib.redirect_stdout(sys.stderr):
                out = gp.Graphpope(d, "flickr", "geodesic", "stochastic", k, None, 2)
            if it % 8 == 0:
                want = oracle.geodesic_features(x, ei, n, d.anchor_nodes)
                assert np.array_equal(out.numpy().view(np.uint32), want.view(np.uint32)), it
            junk = torch.rand(1 << 22)
            assert torch.equal(junk.to(dev).cpu(), junk)
            del out, junk, d
            if it % 7 == 0:
                gc.collect()
            if it in (calls // 3, calls - 1):
                gc.collect()
                torch.cuda.synchronize()
                marks.append((proc.memory_info().rss, torch.cuda.memory_allocated(), proc.num_fds()))
    finally:
        if old is None:
            os.environ.pop("GRAPHPOPE_HOST_RESULT", None)
        else:
            os.environ["GRAPHPOPE_HOST_RESULT"] = old
        gp.clear_cache()
    (rss0, cuda0, fds0), (rss1, cuda1, fds1) = marks
    assert rss1 - rss0 < 256 << 20 and cuda1 - cuda0 < 64 << 20 and fds1 <= fds0 + 2, marks


@pytest.mark.parametrize("seed", range(max(3, SEEDS // 2)))
def test_fan_out_sampler_on_random_graphs(seed, oracle):
    """The fan-out sampler (main.py:100-116, NeighborSampler(sizes)) on random graphs, seed sets and fan-outs -- host-sized and
    device-extent forms (a reused DeviceBatch: the buffers hold the previous batch) -- against its CPU restatement, bit for bit."""
    from graphpope_amd import engine
    from graphpope_amd.sampler import DeviceBatch, NeighborSampler
    dev = engine.require_gpu()
    rs = np.random.RandomState(11000 + BASE + seed)
    n = int(rs.randint(30, 5000))
    e = int(n * rs.uniform(0.5, 12.0))
    ei = rs.randint(0, n, (2, e)).astype(np.int64)
    if rs.rand() < 0.5:                                               # a few hubs
        ei[0, : e // 5] = rs.randint(0, max(n // 50, 1), e // 5)
    csr = engine.build_csr(torch.as_tensor(ei, device=dev), n)
    rowptr, col = csr.rowptr.cpu().numpy(), csr.col.cpu().numpy()[: csr.num_edges]
    batch = None
    for case in range(4):
        hops = int(rs.randint(1, 4))
        sizes = tuple(int(rs.choice([1, 2, 3, 10, 25])) for _ in range(hops))
        b = int(rs.choice([1, 7, 64, 257, min(n, 1550)]))
        b = min(b, n)
        seeds = rs.choice(n, b, replace=False).astype(np.int64)
        sd = int(rs.randint(0, 2**31)) + (int(rs.randint(0, 2**20)) << 32)
        want_n_id, want = seeds, []
        for hop, size in enumerate(sizes):
            rp, cl, want_n_id = oracle.sample_hop(rowptr, col, want_n_id, size, sd, hop)
            want.append((rp, cl, len(want_n_id)))
        sampler = NeighborSampler(csr.rowptr, csr.col, n, sizes)
        seeds_dev = torch.as_tensor(seeds, device=dev)
        n_id, adjs = sampler.sample(seeds_dev, seed=sd)
        assert np.array_equal(n_id.cpu().numpy(), want_n_id), (seed, case)
        for adj, (rp, cl, n_src) in zip(adjs, want[::-1]):
            assert adj.n_src == n_src and np.array_equal(adj.rowptr.cpu().numpy(), rp) and np.array_equal(adj.col.cpu().numpy(), cl), (seed, case)
        if batch is None or batch.n_seeds != b or batch.sizes != list(sizes):
            batch = DeviceBatch(b, sizes, dev)
        for _ in range(2):                                            # the second fill meets the first one's contents
            sampler.sample_device(seeds_dev, seed=sd, out=batch)
            dims = batch.dims.cpu().numpy()
            assert np.array_equal(batch.n_id.cpu().numpy()[: dims[-1, 1]], want_n_id), (seed, case)
            for h, (rp, cl, n_src) in enumerate(want):
                assert dims[h, 1] == n_src and dims[h, 2] == len(cl) and dims[h, 0] == len(rp) - 1
                assert np.array_equal(batch.rowptrs[h].cpu().numpy()[: len(rp)], rp) and np.array_equal(batch.cols[h].cpu().numpy()[: len(cl)], cl), (seed, case, h)


@pytest.mark.parametrize("seed", range(max(3, SEEDS // 2)))
def test_node2vec_embedding_on_random_shapes(seed, oracle):
    """The node2vec-space embedding (utils.py:149-180: distance to the anchors' rows + per-column min-max) on random shapes -- depths on
    both sides of the anchor-resident kernel's limit, anchor counts that leave ragged column groups, feature widths 0 .. 1 028, repeated
    anchors and coincident rows -- against the oracle at the path's stated tolerance (1e-5 absolute on the scaled values)."""
    from graphpope_amd import engine
    dev = engine.require_gpu()
    rs = np.random.RandomState(13000 + BASE + seed)
    for case in range(3):
        n = int(rs.choice([2, 40, 257, 1000, 4099, 20011]))
        d = int(rs.choice([1, 4, 7, 36, 64, 100, 128, 129, 200]))
        k = int(rs.choice([1, 2, 33, 64, 200, 256, 300, 513]))
        f = int(rs.choice([0, 3, 8, 500, 1028]))
        fn = ["distance", "similarity", "euclidean"][int(rs.randint(3))]
        emb = rs.randn(n, d).astype(np.float32)
        if n > 12:
            emb[5] = emb[9]
        anchors = rs.choice(n, k)
        x = rs.rand(n, f).astype(np.float32)
        out = engine.pairwise_features(torch.as_tensor(x, device=dev), torch.as_tensor(emb, device=dev), anchors, fn).cpu().numpy()
        want = oracle.node2vec_features(x, emb, anchors, fn)
        assert out.shape == want.shape and np.array_equal(out[:, :f], x), (seed, case)
        np.testing.assert_allclose(out[:, f:], want[:, f:], rtol=0, atol=1e-5, err_msg=str((seed, case, n, d, k, f, fn)))


@pytest.mark.parametrize("seed", range(max(3, SEEDS // 2)))
def test_sage_conv_on_random_block_shapes(seed, oracle):
    """SAGEConv forward and every gradient (main.py:206: convs[i]((x, x[:n_dst]), adj_t)) on random bipartite block shapes -- the shape
    decides between whole-tile, stream-K, twin and generic GEMM paths, vector and scalar gathers, with and without an input gradient,
    features materialised or read through n_id -- against oracle.sage_conv_torch at the path's tolerances (fwd 1e-4, grads 1e-3 of the
    largest magnitude)."""
    from graphpope_amd import engine
    from graphpope_amd.sage import IndexedFeatures, SAGEConv, SampledAdj
    dev = engine.require_gpu()
    rs = np.random.RandomState(15000 + BASE + seed)

    def close(got, want, rel, what):
        scale = max(float(want.abs().max()), 1e-6)
        err = float((got - want).abs().max())
        assert err <= rel * scale, (seed, what, err, scale, shape)

    for case in range(3):
        n_dst = int(rs.choice([1, 5, 130, 1000, 1550, 4000, 9988]))
        n_src = n_dst + int(rs.choice([0, 1, 300, 5000, 20000]))
        c_in = int(rs.choice([1, 7, 64, 130, 256, 532, 756]))
        c_out = int(rs.choice([1, 3, 32, 256, 257]))
        max_deg = int(rs.choice([0, 1, 4, 10, 25]))
        indexed = bool(rs.randint(2)) and n_dst * c_in < 4_000_000
        shape = (n_dst, n_src, c_in, c_out, max_deg, indexed)
        if n_src * c_in > 12_000_000:
            n_src = n_dst + 300
        deg = rs.randint(0, max_deg + 1, size=n_dst)
        rowptr = torch.tensor(np.concatenate([[0], np.cumsum(deg)]).astype(np.int32))
        col = torch.tensor(rs.randint(0, n_src, size=int(rowptr[-1])).astype(np.int32))
        torch.manual_seed(seed * 7 + case)
        conv = SAGEConv(c_in, c_out).to(dev)
        g = torch.randn(n_dst, c_out)
        adj = SampledAdj(rowptr, col, n_src).to(dev)
        if indexed:
            n_all = n_src + 37
            feats = torch.randn(n_all, c_in)
            n_id = torch.as_tensor(rs.permutation(n_all)[:n_src].astype(np.int64))
            x = feats[n_id]
            out = conv(IndexedFeatures(feats.to(dev), n_id.to(dev)), adj)
            xd = None
        else:
            x = torch.randn(n_src, c_in)
            xd = x.to(dev).requires_grad_(bool(rs.randint(2)))
            out = conv((xd, xd[:n_dst]), adj)
        out.backward(g.to(dev))
        xr = x.clone().requires_grad_(True)
        wl, bl, wr = (p.detach().cpu().clone().requires_grad_(True) for p in (conv.lin_l.weight, conv.lin_l.bias, conv.lin_r.weight))
        ref = oracle.sage_conv_torch(xr, rowptr, col, wl, bl, wr)
        ref.backward(g)
        close(out.detach().cpu(), ref.detach(), 1e-4, "out")
        if xd is not None and xd.requires_grad:
            close(xd.grad.cpu(), xr.grad, 1e-3, "grad_x")
        close(conv.lin_l.weight.grad.cpu(), wl.grad, 1e-3, "grad_w_l")
        close(conv.lin_l.bias.grad.cpu(), bl.grad, 1e-3, "grad_b_l")
        close(conv.lin_r.weight.grad.cpu(), wr.grad, 1e-3, "grad_w_r")
