"""Every instantiation of the level kernel against the oracle on small random graphs (GPU).

The large-graph forms of k_bfs_level (live table read from global memory, with or without its LDS summary; 8-word tiles; tiles walked
inside the wave, in pairs) only run by themselves on graphs of more than 256 Ki nodes, where the oracle needs seconds per case.
POPE_KNOB_LIVE_MODE forces them on small graphs: random directed multigraphs with self-loops, isolated nodes, long paths (deep levels)
and hubs (rows spanning many 256-slot chunks), anchors drawn with repeats, 1 to 1 100 anchors (1 to 18 words per node), on workspaces
that hold the previous case's bytes.  utils.py:64-81 (nx.shortest_path per node and anchor) is what the hop counts must equal.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("seed", range(6))
def test_forced_live_modes_and_tilings_on_random_graphs(seed, oracle):
    from graphpope_amd import _lib, engine
    dev = engine.require_gpu()
    lib = _lib.load()
    rs = np.random.RandomState(1000 + seed)
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
