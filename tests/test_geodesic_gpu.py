"""Parity of the HIP geodesic path (through the C ABI) with the oracle and the reference goldens."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_geodesic_files, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from graphpope_amd import engine
    return engine.require_gpu()


def _run(edge_index, n, anchors, x, dev):
    from graphpope_amd import engine
    ei = torch.as_tensor(np.asarray(edge_index, dtype=np.int64), device=dev)
    csr = engine.build_csr(ei, n)
    hp = engine.bfs(csr, anchors)
    hops = engine.hop_matrix(hp).cpu().numpy()
    xd = torch.as_tensor(x, device=dev)
    out = torch.empty((n, x.shape[1] + len(anchors)), dtype=torch.float32, device=dev)
    engine.finalize(hp.planes, hp.n_hop_bits, n, len(anchors), xd, x.shape[1], out, 0)
    torch.cuda.synchronize()
    e = csr.num_edges
    return hops, out.cpu().numpy(), hp, (csr.rowptr.cpu().numpy(), csr.col.cpu().numpy()[:e], csr.erow.cpu().numpy()[:e])


@pytest.mark.parametrize("path", golden_geodesic_files(), ids=lambda p: os.path.basename(p)[9:-4])
def test_golden_bit_exact(path, dev):
    g = load_golden(path)
    n = int(g["num_nodes"])
    hops, out, hp, _ = _run(g["edge_index"], n, g["anchors"], g["x"], dev)
    assert np.array_equal(hops, g["hops"])
    f = g["x"].shape[1]
    assert np.array_equal(out[:, :f], g["x"])
    assert np.array_equal(out[:, f:].view(np.uint32), g["emb"].view(np.uint32))       # bit-exact float32
    assert hp.max_hop == max(int(g["hops"].max()), 0)


def test_csr_matches_edge_index(dev):
    g = load_golden(os.path.join(GOLDEN, "geodesic_multiloops30.npz"))           # unsorted, loops, repeats
    ei = g["edge_index"].astype(np.int64)
    for edges in (ei, ei[:, np.lexsort((ei[1], ei[0]))]):                       # atomic scatter path, sorted fast path
        _, _, _, (rowptr, col, erow) = _run(edges, 30, g["anchors"], g["x"], dev)
        deg = np.bincount(edges[0], minlength=30)
        assert np.array_equal(np.diff(rowptr), deg) and rowptr[0] == 0
        assert np.array_equal(erow, np.repeat(np.arange(30), deg))
        for v in range(30):
            assert sorted(col[rowptr[v]:rowptr[v + 1]].tolist()) == sorted(edges[1][edges[0] == v].tolist())


@pytest.mark.parametrize("k", [1, 63, 64, 65, 200, 300])
def test_word_tilings_against_oracle(k, dev, oracle):
    from graphpope_amd import synth
    ei, n = synth.rmat(12, edge_factor=6, seed=9)
    anchors = np.random.RandomState(k).choice(np.arange(n), k)
    x = np.random.RandomState(1).rand(n, 6).astype(np.float32)          # F = 6, K odd: scalar store path too
    hops, out, _, _ = _run(ei, n, anchors, x, dev)
    want = oracle.geodesic_hops(ei, n, anchors)
    assert np.array_equal(hops, want)
    assert np.array_equal(out.view(np.uint32), oracle.geodesic_features(x, ei, n, anchors).view(np.uint32))


def test_hub_rows_and_unsorted_edges(dev, oracle):
    """Rows far longer than the per-group limit go through the block sweep; shuffled edge order hits the atomic fill."""
    from graphpope_amd import synth
    ei = synth.powerlaw_graph(20000, 150000, seed=3, alpha=1.0, shift=0.3)
    n = 20000
    assert np.bincount(ei[0]).max() > 4000
    perm = np.random.RandomState(0).permutation(ei.shape[1])
    ei = ei[:, perm]
    anchors = np.random.RandomState(2).choice(np.arange(n), 96)
    x = np.zeros((n, 4), dtype=np.float32)
    hops, _, _, _ = _run(ei, n, anchors, x, dev)
    assert np.array_equal(hops, oracle.geodesic_hops(ei, n, anchors))


def test_deferred_check_rebuilds_unsorted_edge_lists(dev, oracle):
    """defer_check: the sorted fast path is speculative; bfs() picks up the verdict and rebuilds through the counting sort."""
    from graphpope_amd import engine, synth, _lib
    ei, n = synth.rmat(10, edge_factor=8, seed=5)
    ei = ei[:, np.random.RandomState(1).permutation(ei.shape[1])]
    anchors = np.arange(0, 70)
    csr = engine.build_csr(torch.as_tensor(ei, device=dev), n, defer_check=True)
    assert not csr.checked
    hp = engine.bfs(csr, anchors)
    assert csr.checked
    assert np.array_equal(engine.hop_matrix(hp).cpu().numpy(), oracle.geodesic_hops(ei, n, anchors))
    bad = engine.build_csr(torch.tensor([[0, 1], [1, 9]], device=dev), 3, defer_check=True)
    with pytest.raises(_lib.PopeError) as e:
        engine.bfs(bad, [0])
    assert e.value.code == _lib.ERR_INDEX


@pytest.mark.parametrize("name", ["rmat11_seed42", "powerlaw4k_seed42", "path300", "multiloops30", "noedges6", "star701"])
def test_single_call_run_matches_golden(name, dev):
    """pope_geodesic_run: speculative one-sync path (sorted), deep graph (path300: > 16 levels), unsorted input, E = 0."""
    from graphpope_amd import engine
    g = load_golden(os.path.join(GOLDEN, f"geodesic_{name}.npz"))
    n = int(g["num_nodes"])
    ei = torch.as_tensor(g["edge_index"].astype(np.int64), device=dev)
    out, hp = engine.geodesic_run(torch.as_tensor(g["x"], device=dev), ei, n, g["anchors"])
    torch.cuda.synchronize()
    f = g["x"].shape[1]
    assert np.array_equal(out.cpu().numpy()[:, f:].view(np.uint32), g["emb"].view(np.uint32))
    assert np.array_equal(out.cpu().numpy()[:, :f], g["x"])
    assert np.array_equal(engine.hop_matrix(hp).cpu().numpy(), g["hops"])


@pytest.mark.parametrize("k", [64, 100])
def test_large_rmat_waves_loop_over_chunks(k, dev, oracle):
    """R-MAT scale 18 (~3.8 M CSR slots): more 256-slot chunks than resident waves, hubs spanning hundreds of chunks,
    W = 1 and W = 2 word tiles."""
    from graphpope_amd import engine, synth
    ei, n = synth.rmat(18, edge_factor=8, seed=11)
    assert ei.shape[1] > 2048 * 4 * 256 and np.bincount(ei[0]).max() > 5000
    anchors = np.random.RandomState(k).choice(np.arange(n), k)
    _, hp = engine.geodesic_run(None, torch.as_tensor(ei, device=dev), n, anchors, want_out=False)
    assert np.array_equal(engine.hop_matrix(hp).cpu().numpy(), oracle.geodesic_hops(ei, n, anchors))


@pytest.mark.parametrize("mode,k", [(-1, 70), (2, 70), (1, 70), (-1, 300), (2, 300), (-1, 600), (2, 600), (-1, 1000), (2, 1000)])
def test_graph_too_large_for_the_lds_live_table(mode, k, dev, oracle):
    """More than 256 Ki nodes: the level kernel reads the live-bit table from global memory behind a summary staged in LDS
    (k_bfs_level<WT, 3>, built by k_live_summary between the launches; the default there); mode 2 forces the plain global table
    (k_bfs_level<WT, 2>), mode 1 must fall back to the summary form.  k = 300: the node's whole 8-word row in one gather
    (k_bfs_level<8, ., 0>); k = 600: 12 words = three 4-word tiles walked inside the wave, the first two gathered as a pair; k = 1000: two
    8-word tiles walked inside the wave."""
    from graphpope_amd import engine, synth, _lib
    ei, n = synth.rmat(19, edge_factor=3, seed=23)
    assert n > 256 * 1024
    anchors = np.random.RandomState(5).choice(np.arange(n), k)        # k >= 600: several tiles share the live bits
    lib = _lib.load()
    _lib.check(lib.pope_debug_set(_lib.KNOB_LIVE_MODE, mode))
    try:
        _, hp = engine.geodesic_run(None, torch.as_tensor(ei, device=dev), n, anchors, want_out=False)
    finally:
        lib.pope_debug_set(_lib.KNOB_LIVE_MODE, -1)
    assert np.array_equal(engine.hop_matrix(hp).cpu().numpy(), oracle.geodesic_hops(ei, n, anchors))


def test_repeated_launches_are_deterministic(dev):
    """Chunk-spanning rows are accumulated with atomics and committed a level late: the planes must not depend on timing."""
    from graphpope_amd import engine, synth
    ei, n = synth.flickr_like()
    anchors = synth.seeded_anchors(n, 256, 42)
    eid = torch.as_tensor(ei, device=dev)
    ref = None
    for _ in range(20):
        _, hp = engine.geodesic_run(None, eid, n, anchors, want_out=False)
        got = hp.valid().clone()
        if ref is None:
            ref = got
        assert torch.equal(got, ref)


def test_long_path_needs_more_than_8_hop_bits(dev, oracle):
    n = 1500
    a = np.arange(n - 1)
    ei = np.stack([np.concatenate([a, a + 1]), np.concatenate([a + 1, a])])
    anchors = [0, n - 1, 700]
    hops, out, hp, _ = _run(ei, n, anchors, np.zeros((n, 4), np.float32), dev)
    assert hp.max_hop == n - 1 and hp.n_hop_bits == 11
    assert np.array_equal(hops, oracle.geodesic_hops(ei, n, anchors))
    assert out[0, 4 + 1] == np.float32(1.0 / n)


def test_errors_surface_as_exceptions(dev):
    from graphpope_amd import engine, _lib
    ei = torch.tensor([[0, 1], [1, 7]], device=dev)
    with pytest.raises(_lib.PopeError) as e:
        engine.build_csr(ei, 3)
    assert e.value.code == _lib.ERR_INDEX
    csr = engine.build_csr(torch.tensor([[0, 1], [1, 2]], device=dev), 3)
    with pytest.raises(_lib.PopeError) as e:
        engine.bfs(csr, [3])
    assert e.value.code == _lib.ERR_INDEX


def test_flickr_full_size_properties(dev, oracle):
    """BASELINE config 2 at full size: bit-exact against the oracle + size-independent properties."""
    from graphpope_amd import synth
    ei, n = synth.flickr_like()
    anchors = synth.seeded_anchors(n, 256, 42)
    x = np.random.RandomState(0).rand(n, 500).astype(np.float32)
    hops, out, hp, _ = _run(ei, n, anchors, x, dev)
    # (1) an anchor is at distance 0 from itself; (2) duplicate anchors give identical columns
    assert (hops[anchors, np.arange(256)] == 0).all()
    # (3) triangle property along every edge of a symmetric graph: |h(u) - h(v)| <= 1 where both reachable
    hu, hv = hops[ei[0]], hops[ei[1]]
    both = (hu >= 0) & (hv >= 0)
    assert (np.abs(hu - hv)[both] <= 1).all() and ((hu >= 0) == (hv >= 0)).all()
    # (4) every node at hop h > 0 has a neighbour at hop h - 1
    want = oracle.geodesic_hops(ei, n, anchors)
    assert np.array_equal(hops, want)
    assert np.array_equal(out[:, :500], x)
    assert np.array_equal(out[:, 500:].view(np.uint32), oracle.hops_to_embedding(want).view(np.uint32))


def test_graphpope_entry_point(dev, oracle):
    """The public call: same signature, anchors from the global NumPy RNG, cache returns the same object."""
    from graphpope_amd import utils as gp
    g = load_golden(os.path.join(GOLDEN, "geodesic_rmat11_seed42.npz"))

    class Data:
        pass
    d = Data()
    d.x = torch.as_tensor(g["x"])
    d.edge_index = torch.as_tensor(g["edge_index"].astype(np.int64))
    d.num_nodes = int(g["num_nodes"])
    gp.clear_cache()
    np.random.seed(42)
    out = gp.Graphpope(d, "flickr", "geodesic", "stochastic", 16, None, 2)
    assert out.device.type == "cpu" and out.dtype == torch.float32 and out.is_contiguous()
    assert np.array_equal(np.asarray(d.anchor_nodes), g["anchors"])
    assert np.array_equal(out.numpy()[:, 3:].view(np.uint32), g["emb"].view(np.uint32))
    assert gp.Graphpope(d, "pubmed", "node2vec", "kmeans", 3) is out          # memoised, args ignored
    gp.clear_cache()
    with pytest.raises(KeyError):
        gp.Graphpope(d, "flickr", "hyperbolic", "stochastic", 4)
    with pytest.raises(UnboundLocalError):
        gp.Graphpope(d, "flickr", "geodesic", "no_such_method", 4)
    gp.clear_cache()


def test_random_graphs_property_sweep(dev, oracle):
    """60 small random graphs (directed and symmetric, sorted and shuffled edge lists, isolated nodes, every word tiling):
    hop matrix bit-exact against the oracle.  Sizes straddle the 256-slot chunk and 4-slot lane boundaries."""
    from graphpope_amd import engine
    rs = np.random.RandomState(2024)
    for trial in range(60):
        n = int(rs.choice([2, 3, 5, 17, 64, 65, 200, 257, 1000]))
        e = int(rs.choice([0, 1, 3, 4, 5, 255, 256, 257, 511, 1024, 1030, 4099]))
        ei = rs.randint(0, n, size=(2, e)).astype(np.int64)
        if trial % 3 == 0 and e:
            ei = np.concatenate([ei, ei[::-1]], axis=1)                    # symmetric
        if trial % 2 == 0 and ei.shape[1]:
            ei = ei[:, np.lexsort((ei[1], ei[0]))]                           # sorted by source: fast CSR path
        if trial % 5 == 0 and ei.shape[1] > 300:
            ei[0, :300] = ei[0, 0]                                          # one long row (hub spanning chunks)
            ei = ei[:, np.lexsort((ei[1], ei[0]))]
        k = int(rs.choice([1, 2, 63, 64, 65, 128, 129, 257]))
        anchors = rs.randint(0, n, size=k)
        _, hp = engine.geodesic_run(None, torch.as_tensor(ei, device=dev), n, anchors, want_out=False)
        got = engine.hop_matrix(hp).cpu().numpy()
        want = oracle.geodesic_hops(ei, n, anchors)
        assert np.array_equal(got, want), (trial, n, ei.shape[1], k)


def test_persisted_plane_cache(dev, tmp_path, monkeypatch):
    """GRAPHPOPE_CACHE_DIR: the second process-independent call expands cached planes instead of running the BFS."""
    from graphpope_amd import engine, utils as gp
    g = load_golden(os.path.join(GOLDEN, "geodesic_powerlaw4k_seed42.npz"))

    class Data:
        pass
    d = Data()
    d.x, d.edge_index, d.num_nodes = torch.as_tensor(g["x"]), torch.as_tensor(g["edge_index"].astype(np.int64)), int(g["num_nodes"])
    monkeypatch.setenv("GRAPHPOPE_CACHE_DIR", str(tmp_path))
    outs = []
    for attempt in range(2):
        gp.clear_cache()
        np.random.seed(42)
        if attempt == 1:
            monkeypatch.setattr(engine, "geodesic_run", lambda *a, **k: (_ for _ in ()).throw(AssertionError("BFS ran on a cache hit")))
        outs.append(gp.Graphpope(d, "flickr", "geodesic", "stochastic", 32, None, 2).numpy())
    gp.clear_cache()
    assert len(list(tmp_path.glob("pope_*.npz"))) == 1
    for out in outs:
        assert np.array_equal(out[:, 3:].view(np.uint32), g["emb"].view(np.uint32))
    np.random.seed(7)                                    # other anchors -> other key -> miss (the patched BFS raises)
    with pytest.raises(AssertionError):
        gp.Graphpope(d, "flickr", "geodesic", "stochastic", 32, None, 2)
    gp.clear_cache()


@pytest.mark.parametrize("symmetric", [True, False])
def test_closeness_centrality_matches_networkx_bit_for_bit(symmetric, dev):
    """sampling_method='closeness_centrality' (utils.py:50-54): same float64 scores as NetworkX, hence the same anchors."""
    import networkx as nx
    from graphpope_amd import engine, synth, utils as gp
    ei, n = synth.rmat(9, edge_factor=4, seed=13, symmetric=symmetric)
    g = nx.DiGraph()
    g.add_nodes_from(range(n))
    g.add_edges_from(zip(ei[0].tolist(), ei[1].tolist()))
    want = nx.closeness_centrality(g)
    got = engine.closeness_centrality(torch.as_tensor(ei, device=dev), n, batch=200)       # ragged last batch
    assert np.array_equal(got, np.array([want[v] for v in range(n)]))

    class Data:
        pass
    d = Data()
    d.edge_index, d.num_nodes = torch.as_tensor(ei), n
    ref = list({k: v for k, v in sorted(want.items(), key=lambda item: item[1])}.keys())[-17:]     # the reference's selection
    assert gp.sample_anchor_nodes(d, 17, "closeness_centrality") == ref


def test_closeness_anchors_match_reference_golden(dev):
    """tests/golden/anchors_centrality.npz: the reference's own sample_anchor_nodes(..., 'closeness_centrality')."""
    from graphpope_amd import utils as gp
    g = np.load(os.path.join(GOLDEN, "anchors_centrality.npz"))

    class Data:
        pass
    d = Data()
    d.edge_index, d.num_nodes = torch.as_tensor(g["edge_index"].astype(np.int64)), int(g["num_nodes"])
    assert gp.sample_anchor_nodes(d, 24, "closeness_centrality") == g["closeness_centrality"].tolist()


def test_back_to_back_runs_keep_their_outputs(dev, oracle):
    """pope_geodesic_run returns once the BFS verdict is known, while the expansion may still be running: a second call on
    the same stream and the same (reused) workspace must neither disturb the first call's output nor its own."""
    from graphpope_amd import engine, synth
    ei, n = synth.rmat(13, edge_factor=8, seed=17)
    eid = torch.as_tensor(ei, device=dev)
    x = torch.rand(n, 20, device=dev)
    a1 = np.random.RandomState(1).choice(np.arange(n), 256)
    a2 = np.random.RandomState(2).choice(np.arange(n), 256)
    outs = []
    for rep in range(3):
        for a in (a1, a2):
            outs.append((a, engine.geodesic_run(x, eid, n, a, reuse_workspace=True)[0]))     # no synchronisation in between
    torch.cuda.synchronize()
    want = {id(a): oracle.geodesic_features(x.cpu().numpy(), ei, n, a) for a in (a1, a2)}
    for a, out in outs:
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want[id(a)].view(np.uint32))


def test_speculative_window_follows_the_previous_depth_and_survives_a_wrong_guess(dev, oracle):
    """pope_geodesic_run sizes its speculative run of levels by the depth the previous call with the same (N, E, K) found.
    Three graphs with identical sizes and very different depths, interleaved: every result stays bit-exact whether the guess
    was right (second call on a graph), too shallow (the deep graphs after the shallow one: the call continues on the general
    path) or too deep."""
    from graphpope_amd import engine
    n = 1500
    rs = np.random.RandomState(5)
    path = np.arange(n - 1)
    deep = np.concatenate([np.stack([path, path + 1]), np.stack([path + 1, path])], axis=1)                 # a path: depth up to 1 499
    e = deep.shape[1]
    src = rs.randint(0, n, e // 2); dst = rs.randint(0, n, e // 2)
    shallow = np.concatenate([np.stack([src, dst]), np.stack([dst, src])], axis=1)                         # random: depth ~ 10
    comb = np.arange(0, n - 20, 1)
    medium = np.concatenate([np.stack([comb, comb + 20]), np.stack([comb + 20, comb])], axis=1)            # 20 interleaved paths: depth ~ 74
    medium = np.concatenate([medium, shallow[:, : e - medium.shape[1]]], axis=1)
    graphs = []
    for g in (shallow, deep, medium):
        assert g.shape[1] == e
        order = np.lexsort((g[1], g[0]))
        graphs.append(np.ascontiguousarray(g[:, order]).astype(np.int64))
    anchors = rs.choice(n, 64, replace=False)
    x = torch.rand(n, 8, device=dev)
    want = [oracle.geodesic_features(x.cpu().numpy(), g, n, anchors) for g in graphs]
    depths = []
    for which in (0, 0, 1, 1, 0, 2, 2, 1, 0):
        out, hp = engine.geodesic_run(x, torch.as_tensor(graphs[which], device=dev), n, anchors)
        depths.append(hp.max_hop)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want[which].view(np.uint32)), which
    assert depths[0] == depths[1] == depths[4] == depths[8] and depths[2] == depths[3] == depths[7] and depths[2] > depths[5] == depths[6] > depths[0] > 12


def test_runs_on_a_side_stream(dev, oracle):
    """Everything is enqueued on the caller's current stream (also the early verdict read and the asynchronous expansion)."""
    from graphpope_amd import engine, synth
    ei, n = synth.rmat(12, edge_factor=8, seed=29)
    eid = torch.as_tensor(ei, device=dev)
    x = torch.rand(n, 12, device=dev)
    anchors = np.random.RandomState(4).choice(np.arange(n), 100)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        out, hp = engine.geodesic_run(x, eid, n, anchors)
        hops = engine.hop_matrix(hp)
    side.synchronize()
    assert np.array_equal(hops.cpu().numpy(), oracle.geodesic_hops(ei, n, anchors))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), oracle.geodesic_features(x.cpu().numpy(), ei, n, anchors).view(np.uint32))


@pytest.mark.parametrize("symmetric", [True, False])
def test_pagerank_matches_networkx_bit_for_bit(symmetric, dev):
    """sampling_method='pagerank' (utils.py:26-30): the SpMV power iteration on the device gives NetworkX's float64 scores
    bit for bit (shuffled edge list with repeated edges and dangling nodes), hence the reference's anchors."""
    import networkx as nx
    from graphpope_amd import engine, synth, utils as gp
    ei, n = synth.rmat(11, edge_factor=4, seed=21, symmetric=symmetric)
    ei = np.concatenate([ei, ei[:, :500]], axis=1)
    ei = ei[:, np.random.RandomState(3).permutation(ei.shape[1])]
    g = nx.DiGraph()
    g.add_nodes_from(range(n))
    g.add_edges_from(zip(ei[0].tolist(), ei[1].tolist()))
    want = nx.pagerank(g)
    got = engine.pagerank(torch.as_tensor(ei, device=dev), n)
    assert np.array_equal(got, np.array([want[v] for v in range(n)]))

    class Data:
        pass
    d = Data()
    d.edge_index, d.num_nodes = torch.as_tensor(ei), n
    ref = list({k: v for k, v in sorted(want.items(), key=lambda item: item[1])}.keys())[-33:]      # the reference's selection
    assert gp.sample_anchor_nodes(d, 33, "pagerank") == ref


def test_pagerank_anchors_match_reference_golden_and_flickr_size(dev):
    """tests/golden/anchors_centrality.npz holds the reference's own sample_anchor_nodes(..., 'pagerank'); at Flickr size the
    GPU scores are compared with NetworkX run here."""
    import networkx as nx
    from graphpope_amd import engine, synth, utils as gp
    g = np.load(os.path.join(GOLDEN, "anchors_centrality.npz"))

    class Data:
        pass
    d = Data()
    d.edge_index, d.num_nodes = torch.as_tensor(g["edge_index"].astype(np.int64)), int(g["num_nodes"])
    assert gp.sample_anchor_nodes(d, 24, "pagerank") == g["pagerank"].tolist()
    ei, n = synth.flickr_like()
    G = nx.DiGraph()
    G.add_nodes_from(range(n))
    G.add_edges_from(zip(ei[0].tolist(), ei[1].tolist()))
    want = nx.pagerank(G)
    got = engine.pagerank(torch.as_tensor(ei, device=dev), n)
    assert np.array_equal(got, np.array([want[v] for v in range(n)]))


def test_general_csr_build_is_deterministic(dev):
    """A shuffled edge list goes through the counting path; rows are then sorted by target, so two builds give identical
    arrays (the atomic cursors alone would not) and the fan-out sampler draws the same batch for the same seed."""
    from graphpope_amd import engine, synth
    from graphpope_amd.sampler import NeighborSampler
    ei, n = synth.rmat(13, edge_factor=8, seed=3)
    ei = ei[:, np.random.RandomState(0).permutation(ei.shape[1])]
    eid = torch.as_tensor(ei, device=dev)
    a, b = engine.build_csr(eid, n), engine.build_csr(eid, n)
    e = ei.shape[1]
    assert torch.equal(a.rowptr, b.rowptr) and torch.equal(a.col[:e], b.col[:e]) and torch.equal(a.erow[:e], b.erow[:e])
    col, rowptr = a.col[:e].cpu().numpy(), a.rowptr.cpu().numpy()
    for v in np.random.RandomState(1).choice(n, 200):
        assert np.array_equal(col[rowptr[v]:rowptr[v + 1]], np.sort(ei[1][ei[0] == v]))
    seeds = torch.arange(0, 300, device=dev)
    s1 = NeighborSampler(a.rowptr, a.col, n, (5, 3)).sample(seeds, seed=9)
    s2 = NeighborSampler(b.rowptr, b.col, n, (5, 3)).sample(seeds, seed=9)
    assert torch.equal(s1[0], s2[0]) and all(torch.equal(x.col, y.col) for x, y in zip(s1[1], s2[1]))


@pytest.mark.parametrize("mode,refuse,transport", [("ring", 0, "codes"), ("ring", 0, "float"), ("ring", 3, "codes"), ("ring", 2, "float"),
                                                   ("ring", 6, "float"), ("staged", 0, "codes")],
                         ids=["ring_codes", "ring_float", "nothing_can_be_pinned", "ring_refused_bounce", "ring_and_bounce_refused", "staged"])
def test_host_to_host_call_returns_a_pageable_tensor_in_every_result_mode(mode, refuse, transport, dev, oracle, monkeypatch):
    """utils.py:129-147 from CPU tensors to a CPU tensor, at a size that takes the chunked paths (38 MB result: 4 chunks
    through the 3-slot pinned ring): ordinary pageable memory like the reference's torch.cat, bit-exact, in the default ring
    mode, and when the runtime refuses to register the caller's pages (POPE_KNOB_FAIL_HOST_REGISTER bit 0: edge_index then goes
    through pinned staging), to allocate the pinned ring (bit 1: no byte transport, float columns through the 4 MB bounce
    buffer) or the bounce buffer as well (bit 2: a blocking copy by the runtime).  ``staged``: the caller's edge_index pages are
    never registered either."""
    from graphpope_amd import _lib, synth, utils as gp
    lib = _lib.load()
    ei, n = synth.rmat(15, edge_factor=8, seed=5)
    f, k = 101, 192                                           # odd F: rows of 1172 bytes, chunks end inside pages
    x = torch.rand(n, f, generator=torch.Generator().manual_seed(3))

    class Data:
        pass
    d = Data()
    d.x, d.edge_index, d.num_nodes = x, torch.as_tensor(ei), n
    monkeypatch.setenv("GRAPHPOPE_HOST_RESULT", mode)
    monkeypatch.setenv("GRAPHPOPE_HOST_TRANSPORT", transport)       # byte codes only travel through the ring; without it: floats
    lib.pope_debug_set(_lib.KNOB_FAIL_HOST_REGISTER, refuse)
    try:
        for _ in range(2):                                    # the ring is reused by the second call
            gp.clear_cache()
            np.random.seed(7)
            out = gp.Graphpope(d, "flickr", "geodesic", "stochastic", k, None, 2)
            assert out.device.type == "cpu" and out.is_contiguous() and not out.is_pinned()
            want = oracle.geodesic_features(x.numpy(), ei, n, d.anchor_nodes)
            assert np.array_equal(out.numpy().view(np.uint32), want.view(np.uint32))
    finally:
        lib.pope_debug_set(_lib.KNOB_FAIL_HOST_REGISTER, 0)
        gp.clear_cache()
    assert not d.edge_index.is_pinned()                       # the caller's tensor is released again


def test_host_to_host_call_rejects_an_unknown_result_mode(dev, monkeypatch):
    from graphpope_amd import synth, utils as gp
    ei, n = synth.rmat(8, edge_factor=4, seed=1)

    class Data:
        pass
    d = Data()
    d.x, d.edge_index, d.num_nodes = torch.rand(n, 3), torch.as_tensor(ei), n
    monkeypatch.setenv("GRAPHPOPE_HOST_RESULT", "mapped")
    gp.clear_cache()
    with pytest.raises(ValueError, match="GRAPHPOPE_HOST_RESULT"):
        gp.Graphpope(d, "flickr", "geodesic", "stochastic", 8, None, 2)
    gp.clear_cache()


@pytest.mark.parametrize("refuse", [0, 2, 6], ids=["ring", "bounce", "runtime_copy"])
def test_host_result_assembly_shapes(refuse, dev):
    """pope_assemble_host_result on its own: no feature columns, one chunk, more chunks than rows, a strided x, and a result
    of seven ring chunks with rows of 132 bytes (the ring wraps twice, chunks end inside rows' cache lines) -- through the pinned
    ring, through the 4 MB bounce buffer (the ring refused) and by the runtime's blocking copy (both refused); into heap tensors
    and into page-aligned anonymous mappings like the ones Graphpope() returns.  No page of a result is ever registered."""
    from graphpope_amd import _lib, engine
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    lib.pope_debug_set(_lib.KNOB_FAIL_HOST_REGISTER, refuse)
    try:
        for n, f, k, chunks in ((5, 3, 4, 8), (40000, 0, 64, 8), (70001, 40, 36, 3), (3000, 700, 8, 0), (400003, 7, 33, 8)):
            emb = torch.rand(n, k, generator=g).to(dev)
            xw = torch.rand(n, f + 5, generator=g)
            x = xw[:, :f]                                     # row pitch larger than the row
            for out in (torch.full((n, f + k), -1.0), engine.host_result_tensor(n, f + k)):
                out.fill_(-1.0)
                engine.assemble_host_result(x if f else None, emb, out, f, threads=4, chunks=chunks)
                assert torch.equal(out[:, :f], x) and torch.equal(out[:, f:], emb.cpu())
                del out
        emb = torch.rand(50000, 80, generator=g).to(dev)[:, :48]      # a pitched device embedding (320-byte rows, 192 used)
        out = torch.full((50000, 48), -1.0)
        engine.assemble_host_result(None, emb, out, 0, threads=3, chunks=2)
        assert torch.equal(out, emb.cpu())
    finally:
        lib.pope_debug_set(_lib.KNOB_FAIL_HOST_REGISTER, 0)


def test_host_result_rows_wider_than_the_ring_and_the_bounce_buffer(dev):
    """Rows of 9 MB: wider than a slot of the pinned ring (8 MB) and than the bounce buffer (4 MB) -- the columns travel through
    the bounce buffer in column pieces (round 5; they used to fall through to a copy by the runtime), next to 12 feature bytes."""
    from graphpope_amd import engine
    g = torch.Generator().manual_seed(1)
    n, f, k = 3, 3, (9 << 20) // 4 + 5
    emb = torch.rand(n, k, generator=g).to(dev)
    x = torch.rand(n, f, generator=g)
    out = torch.full((n, f + k), -1.0)
    engine.assemble_host_result(x, emb, out, f, threads=2, chunks=0)
    assert torch.equal(out[:, :f], x) and torch.equal(out[:, f:], emb.cpu())


@pytest.mark.parametrize("k", [1, 37, 64, 200, 1024])
def test_hop_codes_are_the_hop_matrix_plus_one(k, dev, oracle):
    """pope_geodesic_hop_codes, the transport form of the embedding: code 0 = no path, hops + 1 otherwise, and the table
    holds exactly the floats the finalise kernel writes (utils.py:73) -- for widths that are and are not multiples of 4."""
    from graphpope_amd import engine, synth
    ei, n = synth.rmat(11, edge_factor=3, seed=2)                 # sparse enough to leave some pairs without a path
    anchors = synth.seeded_anchors(n, k, 5)
    ei_dev = torch.as_tensor(ei).to(dev)
    emb, hp = engine.geodesic_run(None, ei_dev, n, anchors)
    codes, lut = engine.hop_codes(hp)
    hops = oracle.geodesic_hops(ei, n, anchors)
    assert (hops < 0).any() and hops.max() >= 2
    assert np.array_equal(codes.cpu().numpy().astype(np.int64), hops.astype(np.int64) + 1)
    want_lut = np.zeros(256, dtype=np.float32)
    want_lut[1:] = np.float32(1.0) / np.arange(1, 256, dtype=np.float32)
    assert np.array_equal(lut.cpu().numpy().view(np.uint32), want_lut.view(np.uint32))
    assert np.array_equal(lut.cpu().numpy()[codes.cpu().numpy()].view(np.uint32), emb.cpu().numpy().view(np.uint32))


def test_hop_codes_refuse_hop_counts_that_do_not_fit_a_byte(dev):
    from graphpope_amd import _lib, engine
    n = 300                                                       # a path: node 299 is 299 hops from anchor 0
    ei = np.stack([np.arange(n - 1), np.arange(1, n)]).astype(np.int64)
    ei = np.concatenate([ei, ei[::-1]], axis=1)
    _, hp = engine.geodesic_run(None, torch.as_tensor(ei).to(dev), n, [0, 150], want_out=False)
    assert hp.max_hop == 299
    with pytest.raises(_lib.PopeError, match="254"):
        engine.hop_codes(hp)


@pytest.mark.parametrize("transport", ["codes", "float"])
def test_host_to_host_call_on_a_graph_deeper_than_a_byte(transport, dev, oracle, monkeypatch):
    """299 hops: the byte transport does not apply and the call sends the float columns instead -- same result."""
    from graphpope_amd import utils as gp
    n = 300
    ei = np.stack([np.arange(n - 1), np.arange(1, n)]).astype(np.int64)
    ei = np.concatenate([ei, ei[::-1]], axis=1)

    class Data:
        pass
    d = Data()
    d.x, d.edge_index, d.num_nodes = torch.rand(n, 5), torch.as_tensor(ei), n
    monkeypatch.setenv("GRAPHPOPE_HOST_TRANSPORT", transport)
    gp.clear_cache()
    np.random.seed(1)
    out = gp.Graphpope(d, "flickr", "geodesic", "stochastic", 12, None, 2)
    gp.clear_cache()
    want = oracle.geodesic_features(d.x.numpy(), ei, n, d.anchor_nodes)
    assert np.array_equal(out.numpy().view(np.uint32), want.view(np.uint32))


def test_host_result_assembly_from_codes(dev):
    """pope_assemble_finish_codes on its own: every code value, widths that are not multiples of 8, feature widths that put
    the embedding columns at every alignment, a strided code matrix, and a result of four ring chunks."""
    from graphpope_amd import engine
    g = torch.Generator().manual_seed(1)
    lut = torch.rand(256, generator=g).to(dev)
    for n, f, k in ((7, 0, 1), (5000, 3, 37), (5000, 1, 8), (5000, 2, 255), (40000, 6, 64), (300007, 5, 100), (30011, 4, 1024)):
        wide = torch.randint(0, 256, (n, k + 3), generator=g, dtype=torch.uint8).to(dev)
        codes = wide[:, :k]                                   # row pitch larger than the row
        x = torch.rand(n, f, generator=g)
        out = torch.full((n, f + k), -1.0)
        with engine.HostAssembly(x if f else None, out, f, threads=5) as asm:
            asm.finish_codes(codes, lut)
        assert torch.equal(out[:, :f], x)
        assert torch.equal(out[:, f:], lut.cpu()[codes.cpu().long()])


def test_concurrent_host_assemblies_take_turns_on_the_ring(dev):
    """Two host threads, each on its own stream, assembling different results at the same time (ctypes releases the GIL):
    the process-wide pinned ring is one resource -- pope_assemble_finish holds its mutex for the D2H phase -- so the calls
    interleave without mixing chunks; float and byte transports at once."""
    import threading
    from graphpope_amd import engine
    g = torch.Generator().manual_seed(4)
    lut = torch.rand(256, generator=g).to(dev)
    jobs = []
    for n, f, k, coded in ((120001, 9, 96, False), (90007, 4, 200, True)):
        x = torch.rand(n, f, generator=g)
        emb = (torch.randint(0, 256, (n, k), generator=g, dtype=torch.uint8) if coded else torch.rand(n, k, generator=g)).to(dev)
        jobs.append((x, emb, coded, f))
    torch.cuda.synchronize()
    errors, results = [], {}

    def worker(idx):
        try:
            x, emb, coded, f = jobs[idx]
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                for it in range(4):
                    out = torch.full((x.shape[0], f + emb.shape[1]), -1.0)
                    with engine.HostAssembly(x, out, f, threads=3) as asm:
                        asm.finish_codes(emb, lut) if coded else asm.finish(emb)
                    results[(idx, it)] = out
        except Exception as exc:                                  # surfaced in the main thread below
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors and all(not t.is_alive() for t in threads), errors
    for (idx, it), out in results.items():
        x, emb, coded, f = jobs[idx]
        want = lut.cpu()[emb.cpu().long()] if coded else emb.cpu()
        assert torch.equal(out[:, :f], x) and torch.equal(out[:, f:], want), (idx, it)
    assert len(results) == 8


def test_odd_plane_length_is_cleared_to_the_last_word(dev):
    """N * W odd (N = 201, one 64-anchor word per node): the (1 + 4) eagerly cleared planes are 8 mod 16 bytes long, so a clear in
    16-byte units stops one word short -- the hop-bit-3 word of node N - 1.  The reference's golden (utils.py:64-81;
    depth 200, node N - 1 is an anchor: hop 0) through every clearing path, each on memory filled with 0xFF: pope_geodesic_run with
    the merged prepare launch and with the separate launches (reused workspace), and pope_geodesic_bfs on poisoned planes and
    scratch (k_zero; the deep planes' clears start 8 bytes off a 16-byte boundary)."""
    import ctypes
    from graphpope_amd import _lib, engine
    lib = _lib.load()
    g = load_golden(os.path.join(GOLDEN, "geodesic_path201_odd.npz"))
    n, anchors = int(g["num_nodes"]), np.ascontiguousarray(g["anchors"], dtype=np.int64)
    assert n % 2 == 1 and len(anchors) <= 64 and int(g["hops"].max()) >= 8
    ei = torch.as_tensor(g["edge_index"].astype(np.int64), device=dev)
    try:
        for merge in (1, 0):
            lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, merge)
            engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)     # creates the workspace of this size
            for _ in range(2):
                for ws in engine._WORKSPACE.values():
                    ws.fill_(0xFF)
                _, hp = engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)
                assert np.array_equal(engine.hop_matrix(hp).cpu().numpy(), g["hops"]), merge
    finally:
        lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 1)
    csr = engine.build_csr(ei, n)
    k, cap = len(anchors), 8
    planes = torch.full((cap + 1, n, lib.pope_words(k)), -1, dtype=torch.int64, device=dev)
    scratch = torch.full((lib.pope_bfs_scratch_bytes(n, csr.num_edges, k),), 0xFF, dtype=torch.uint8, device=dev)
    max_hop, bits = ctypes.c_int32(0), ctypes.c_int32(0)
    _lib.check(lib.pope_geodesic_bfs(_lib.ptr(csr.rowptr), _lib.ptr(csr.col), _lib.ptr(csr.erow), _lib.ptr(csr.aux), n, csr.num_edges,
                                     ctypes.c_void_p(anchors.ctypes.data), k, _lib.ptr(planes), cap, _lib.ptr(scratch), scratch.numel(),
                                     ctypes.byref(max_hop), ctypes.byref(bits), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    hp = engine.HopPlanes(planes, int(bits.value), int(max_hop.value), n, k)
    assert np.array_equal(engine.hop_matrix(hp).cpu().numpy(), g["hops"])


def test_one_prepare_launch_gives_the_same_bits(dev, oracle):
    """POPE_KNOB_PREPARE_MERGE (the default): pope_geodesic_run clears and seeds the BFS state and builds the speculative CSR as two roles of ONE
    launch (k_prepare: the anchors by value, every seeded word written by the block that zeroed it, the CSR status word tagged with
    the call's epoch instead of zeroed).  Against the reference's goldens (utils.py:64-81, 129-135) on ONE reused workspace -- so
    each call meets the previous call's status word -- with the workspace overwritten by 0xFF and by random bytes in between; the
    unsorted list (fallback to the counting sort), E = 0 and more than 256 anchors (the separate launches) included; a node id out of
    range still fails; at Flickr size with 256 anchors (duplicates among them) the planes equal the separate launches' bit for bit."""
    from graphpope_amd import _lib, engine, synth
    lib = _lib.load()
    lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 1)
    try:
        names = ["rmat11_seed42", "multiloops30", "powerlaw4k_seed42", "path300", "noedges6", "star701", "rmat11_seed42", "multiloops30"]
        for rnd, name in enumerate(names):
            g = load_golden(os.path.join(GOLDEN, f"geodesic_{name}.npz"))
            n = int(g["num_nodes"])
            ei = torch.as_tensor(g["edge_index"].astype(np.int64), device=dev)
            for ws in engine._WORKSPACE.values():
                if rnd % 3 == 1:
                    ws.fill_(0xFF)
                elif rnd % 3 == 2:
                    ws.copy_(torch.randint(0, 256, ws.shape, dtype=torch.uint8, device=ws.device))
            for _ in range(2):
                out, hp = engine.geodesic_run(torch.as_tensor(g["x"], device=dev), ei, n, g["anchors"], reuse_workspace=True)
                f = g["x"].shape[1]
                assert np.array_equal(out.cpu().numpy()[:, f:].view(np.uint32), g["emb"].view(np.uint32)), name
                assert np.array_equal(engine.hop_matrix(hp).cpu().numpy(), g["hops"]), name
        with pytest.raises(_lib.PopeError) as e:
            engine.geodesic_run(None, torch.tensor([[0, 1, 2, 3], [1, 2, 9, 0]], device=dev), 4, [0, 1], want_out=False, reuse_workspace=True)
        assert e.value.code == _lib.ERR_INDEX
        ei_np, n = synth.flickr_like()
        eid = torch.as_tensor(ei_np, device=dev)
        for k in (256, 300, 7):                                     # 300: the separate launches; all with repeated anchors
            anchors = np.random.RandomState(k).choice(np.arange(n), k)
            anchors[-1] = anchors[0]
            lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 0)
            want = engine.geodesic_run(None, eid, n, anchors, want_out=False)[1].valid().clone()
            lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 1)
            for _ in range(3):
                got = engine.geodesic_run(None, eid, n, anchors, want_out=False, reuse_workspace=True)[1].valid()
                assert torch.equal(got, want), k
        want_hops = oracle.geodesic_hops(ei_np, n, anchors)
        assert np.array_equal(engine.hop_matrix(engine.geodesic_run(None, eid, n, anchors, want_out=False)[1]).cpu().numpy(), want_hops)
    finally:
        lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 1)


def test_status_word_of_the_prepare_launch_is_not_a_value_old_index_data_can_hold(dev):
    """The merged prepare launch does not zero the CSR status word, it tags it; the tag must be a value nothing else stores there.
    Round 4 counted its tags 1, 2, 3, ...: (tag << 3) | flags is then a small integer, and a workspace that had held index arrays (node
    ids, row offsets: another graph's CSR at the same address) made a clean edge list fail with "node id outside [0, N)".  In a FRESH
    process (the tag sequence starts over) every call i finds its workspace filled with the int32 the old scheme would have read as
    call i's raised flag, and with other small integers: all must succeed, planes equal to the first call's."""
    import subprocess
    import sys
    code = """
import numpy as np, torch
from graphpope_amd import engine, synth
dev = engine.require_gpu()
ei_np, n = synth.rmat(10, edge_factor=8, seed=3)
ei = torch.as_tensor(ei_np, device=dev)
anchors = np.random.RandomState(0).choice(n, 100)
_, hp = engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)
want = hp.valid().clone()
for i in range(1, 41):
    for ws in engine._WORKSPACE.values():
        w32 = ws[: ws.numel() // 4 * 4].view(torch.int32)
        if i % 2:
            w32.fill_(((i + 1) << 3) | 1)        # call i + 1 of the process, flag 1 (node id out of range) under the counting tags
        else:
            w32.copy_(torch.randint(0, 512, w32.shape, dtype=torch.int32, device=w32.device))
    _, hp = engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)
    assert torch.equal(hp.valid(), want), i
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("f,k,shards", [(0, 64, 1), (20, 100, 1), (300, 256, 1), (500, 256, 1), (700, 512, 1), (36, 1024, 1), (1100, 64, 1),
                                         (0, 128, 3), (40, 256, 4), (8, 128, 8), (0, 256, 5), (0, 256, 8), (500, 1024, 3), (12, 1020, 1),
                                         (0, 64, 8), (0, 64, 4), (0, 128, 8), (0, 512, 2), (0, 1024, 2), (0, 2048, 2), (16, 64, 8)])
def test_pipelined_finalise_kernel_writes_the_same_bits(dev, f, k, shards):
    """The finalise kernel of round 4 (every load of a row in flight at once, the next row requested before this one is stored, rows
    dealt round-robin to the waves; utils.py:73, 129-135) against the round 1-3 kernel (POPE_KNOB_FINALIZE_VARIANT 7) on random
    planes: one and several shards, feature widths and anchor counts on both sides of every instance's limits (and beyond them,
    where the old kernel takes over) -- the same bits, NaN-poisoned outputs; and the library's own name for the kernel it picks
    (pope_finalize_kernel_name, what bench.py labels its roofline entry with) follows the shape.  Several shards of 64 to 1 024 anchors:
    the table kernel takes a batch per (shard, block of rows), 12 = the flat order it had before."""
    from graphpope_amd import _lib, engine
    lib = _lib.load()
    n, bits = 3001, 4
    g = torch.Generator().manual_seed(f + 7 * k + shards)
    w = lib.pope_words(k)
    planes = torch.randint(-2**62, 2**62, (shards, 1 + bits, n, w), generator=g, dtype=torch.int64).to(dev)
    x = torch.rand((n, f), generator=g).to(dev) if f else None
    outs = []
    try:
        for variant in (7, 8, 9, 0, 10, 12):                       # 8: the default; 9: the table kernel with features too; 10: wide rows on the shuffle kernel
            lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 9 if variant == 12 else variant)
            lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 12 if variant == 12 else 11)
            out = torch.full((n, f + shards * k), float("nan"), device=dev)
            if shards == 1:
                engine.finalize(planes[0], bits, n, k, x, f, out, 0)
            else:
                engine.finalize_shards(planes, bits, n, k, x, f, out)
            outs.append(out)
    finally:
        lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 8)          # the defaults again
        lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 11)
    assert not torch.isnan(outs[0]).any()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    import ctypes
    name = ctypes.create_string_buffer(64)
    _lib.check(lib.pope_finalize_kernel_name(n, k, f, 1 if f else 0, shards, name, 64))
    ne = k // 4 * shards
    pow2 = lambda v: v > 0 and v & (v - 1) == 0
    wide = ne > 64 and k % 64 == 0
    want = "k_finalize_fast" if f > 1024 else ("k_finalize_lut" if wide and not f and pow2(k // 64) and pow2(shards) else "k_finalize_wide" if wide else "k_finalize_pipe")
    assert name.value.decode().startswith(want), (name.value, want)
