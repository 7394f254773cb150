"""The CPU oracle against the golden vectors the reference produced (tests/golden/make_goldens.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_geodesic_files, load_golden


@pytest.mark.parametrize("path", golden_geodesic_files(), ids=lambda p: os.path.basename(p)[9:-4])
def test_geodesic_oracle_matches_reference_bit_exact(oracle, path):
    g = load_golden(path)
    n = int(g["num_nodes"])
    hops = oracle.geodesic_hops(g["edge_index"].astype(np.int64), n, g["anchors"])
    assert hops.dtype == np.int32 and hops.shape == g["hops"].shape
    assert np.array_equal(hops, g["hops"])
    emb = oracle.hops_to_embedding(hops)
    assert emb.dtype == np.float32
    assert np.array_equal(emb.view(np.uint32), g["emb"].view(np.uint32))          # bit-exact f32
    out = oracle.concat_into_features(g["x"], emb)
    assert out.shape == (n, g["x"].shape[1] + len(g["anchors"]))
    assert np.array_equal(out[:, :g["x"].shape[1]], g["x"]) and np.array_equal(out[:, g["x"].shape[1]:], g["emb"])


def test_goldens_cover_the_edge_cases():
    names = {os.path.basename(p)[9:-4] for p in golden_geodesic_files()}
    assert {"digraph5", "path300", "oneway40", "star701", "components60", "multiloops30", "rmat8_k130",
            "rmat9_directed", "rmat11_seed42", "powerlaw4k_seed42", "sparse1500_seed7", "noedges6"} <= names
    assert load_golden(os.path.join(GOLDEN, "geodesic_path300.npz"))["hops"].max() == 299      # > 8 bits
    g = load_golden(os.path.join(GOLDEN, "geodesic_rmat8_k130.npz"))
    assert len(g["anchors"]) == 130 and len(np.unique(g["anchors"])) < 130                       # duplicates kept


def test_f32_division_equals_reference_double_rounding():
    # utils.py:73 divides in float64 and torch.as_tensor rounds to float32; the device computes
    # 1.0f / (float)(h + 1).  The two agree for every hop count a graph with < 2^24 nodes can produce.
    h1 = np.arange(1, 1 << 24, dtype=np.int64)
    via_f64 = (1.0 / h1.astype(np.float64)).astype(np.float32)
    via_f32 = np.float32(1.0) / h1.astype(np.float32)
    assert np.array_equal(via_f64.view(np.uint32), via_f32.view(np.uint32))


def test_oracle_rejects_out_of_range_ids(oracle):
    with pytest.raises(IndexError):
        oracle.geodesic_hops(np.array([[0], [5]]), 3, [0])
    with pytest.raises(IndexError):
        oracle.geodesic_hops(np.array([[0], [1]]), 3, [3])


def test_oracle_pairs_networkx_agrees(oracle):
    g = load_golden(os.path.join(GOLDEN, "geodesic_rmat9_directed.npz"))
    nodes = np.arange(0, 512, 37)
    ref = oracle.geodesic_pairs_networkx(g["edge_index"].astype(np.int64), 512, g["anchors"][:6], nodes)
    assert np.array_equal(ref, g["emb"][nodes][:, :6])


def test_seeded_anchor_draw_matches_reference(oracle):
    from graphpope_amd import synth
    with np.load(os.path.join(GOLDEN, "anchors_stochastic.npz")) as z:
        for key in z.files:
            n, k, s = (int(t[1:]) for t in key.split("_"))
            np.random.seed(s)
            assert np.array_equal(oracle.sample_anchor_nodes_stochastic(n, k), z[key])
            assert np.array_equal(synth.seeded_anchors(n, k, s), z[key])
    assert synth.seeded_anchors(89250, 256, 42)[:8].tolist() == [15795, 860, 76820, 54886, 6265, 82386, 37194, 87498]


@pytest.mark.parametrize("family", ["randn2048", "small96", "const40"])
@pytest.mark.parametrize("fn", ["distance", "similarity", "euclidean"])
def test_node2vec_oracle_matches_reference(oracle, family, fn):
    g = load_golden(os.path.join(GOLDEN, f"node2vec_{family}.npz"))
    out = oracle.node2vec_features(g["x"], g["emb"], g["anchors"], fn)
    f = g["x"].shape[1]
    assert out.dtype == np.float32 and np.array_equal(out[:, :f], g["x"])
    # tolerance: SURVEY.md §8c, 1e-5 abs on the min-max scaled embedding
    np.testing.assert_allclose(out[:, f:], g[f"scaled_{fn}"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("fn", ["distance", "similarity", "euclidean"])
def test_node2vec_kmeans_anchors_oracle_matches_reference(oracle, fn):
    """utils.py:168-170: K-means centres as anchors (golden = the reference's own Graphpope(..., 'kmeans', ...))."""
    g = load_golden(os.path.join(GOLDEN, "node2vec_kmeans512.npz"))
    out = oracle.node2vec_features(g["x"], g["emb"], None, fn, anchor_embeddings=g["centres"])
    f = g["x"].shape[1]
    np.testing.assert_allclose(out[:, f:], g[f"scaled_{fn}"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("name", ["node2vec_kmeans_k80.npz", "node2vec_kmeans_overlap256.npz"])
def test_node2vec_larger_kmeans_goldens_pin_the_oracle(oracle, name):
    """The two larger reference-held K-means cases (80 separated blobs; 256 clusters on an overlapping N(0, 1) table, the
    kind of data the reference really clusters): given the reference's centres, the oracle's distances + min-max scaling
    reproduce the reference's columns."""
    g = load_golden(os.path.join(GOLDEN, name))
    out = oracle.node2vec_features(g["x"], g["emb"], None, "euclidean", anchor_embeddings=g["centres"])
    f = g["x"].shape[1]
    np.testing.assert_allclose(out[:, f:], g["scaled_euclidean"], rtol=0, atol=1e-5)


def test_node2vec_unknown_distance_function_is_keyerror(oracle):
    with pytest.raises(KeyError):
        oracle.pairwise(np.zeros((2, 2), np.float32), np.zeros((1, 2), np.float32), "manhattan")


def test_sampler_restatement_properties(oracle):
    """The CPU restatement of the fan-out sampler: distinct true neighbours, whole rows below the fan-out, targets first."""
    from graphpope_amd import synth
    ei = synth.powerlaw_graph(800, 4000, seed=3, alpha=0.9, shift=0.8)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei[0], minlength=800))])
    col = ei[1]
    seeds = np.arange(0, 200)
    rp, cl, n_id = oracle.sample_hop(rowptr, col, seeds, 5, seed=42, hop=0)
    assert np.array_equal(n_id[:200], seeds) and len(np.unique(n_id)) == len(n_id) and cl.max() < len(n_id)
    for i, g in enumerate(seeds):
        true = col[rowptr[g]:rowptr[g + 1]]
        got = n_id[cl[rp[i]:rp[i + 1]]]
        assert len(got) == min(len(true), 5) and len(set(got.tolist())) == len(got) and set(got.tolist()) <= set(true.tolist())
    for d in (1, 2, 3, 17, 64, 65, 1000):                     # the keyed Feistel map is a permutation of [0, d)
        assert sorted(oracle.feistel_perm(i, d, 0xC0FFEE) for i in range(d)) == list(range(d))
    assert oracle.sample_hop(rowptr, col, seeds, 5, 42, 0)[1].tolist() == cl.tolist()            # pure function of its arguments
    assert oracle.sample_hop(rowptr, col, seeds, 5, 43, 0)[1].tolist() != cl.tolist()


@pytest.mark.parametrize("method", ["betweenness_centrality", "degree_centrality", "eigenvector_centrality",
                                    "clustering_coefficient"])
def test_biased_anchor_selection_matches_reference(method):
    """utils.py:32-60 (host rankings): same anchors, same order as the reference's own sample_anchor_nodes
    (tests/golden/anchors_centrality.npz).  closeness_centrality and pagerank run on the GPU: tests/test_geodesic_gpu.py."""
    import torch
    from graphpope_amd import utils as gp
    g = np.load(os.path.join(GOLDEN, "anchors_centrality.npz"))

    class Data:
        pass
    d = Data()
    d.edge_index, d.num_nodes = torch.as_tensor(g["edge_index"].astype(np.int64)), int(g["num_nodes"])
    got = gp.sample_anchor_nodes(d, 24, method)
    assert [int(v) for v in got] == g[method].tolist()


def test_pool_restatement_of_the_reference_loop(oracle):
    """oracle.geodesic_pairs_networkx_pool (bench.py's Baseline A: utils.py:92-107 under multiprocessing.Pool) gives the
    reference's own numbers: checked against a golden produced by the reference's utils.py."""
    g = load_golden(os.path.join(GOLDEN, "geodesic_rmat9_directed.npz"))
    n = int(g["num_nodes"])
    nodes = np.arange(0, n, 7)
    emb, tm = oracle.geodesic_pairs_networkx_pool(g["edge_index"], n, g["anchors"], nodes, 3)
    assert np.array_equal(emb.view(np.uint32), g["emb"][nodes].view(np.uint32))
    assert tm["pool_s"] > 0 and tm["graph_build_s"] > 0


def test_pagerank_restatement_matches_networkx_and_the_reference_golden(oracle):
    """oracle.pagerank_scores: bit-identical to nx.pagerank (the SciPy iteration the reference calls, utils.py:28) on directed
    and symmetric graphs with dangling nodes and repeated edges; its last-K selection is the reference's golden."""
    import networkx as nx
    from graphpope_amd import synth
    for seed, sym in ((1, True), (2, False)):
        ei, n = synth.rmat(8, edge_factor=3, seed=seed, symmetric=sym)
        ei = np.concatenate([ei, ei[:, :40]], axis=1)                       # repeated edges collapse in the DiGraph
        g = nx.DiGraph()
        g.add_nodes_from(range(n))
        g.add_edges_from(zip(ei[0].tolist(), ei[1].tolist()))
        want = nx.pagerank(g)
        assert np.array_equal(oracle.pagerank_scores(ei, n), np.array([want[v] for v in range(n)]))
    g = np.load(os.path.join(GOLDEN, "anchors_centrality.npz"))
    score = oracle.pagerank_scores(g["edge_index"].astype(np.int64), int(g["num_nodes"]))
    assert np.argsort(score, kind="stable")[-24:].tolist() == g["pagerank"].tolist()


def test_sage_conv_restatement_against_an_independent_dense_formulation(oracle):
    """oracle.sage_conv_torch (gather + index_add_ + two F.linear, float32) against a second formulation that shares none
    of its steps: the dense float64 operator  out = D^-1 A X W_l^T + b + X_dst W_r^T  with A the [n_dst, n_src] incidence
    count matrix of the sampled block (repeated edges count twice, as a mean over the CSR entries does), and its
    gradients from autograd on that expression.  PyG is absent, so the SAGE oracle cannot be pinned to the reference;
    this at least checks the restatement against something the kernels were not written from."""
    import torch
    rs = np.random.RandomState(0)
    n_src, n_dst, c_in, c_out = 70, 23, 11, 6
    deg = rs.randint(0, 7, n_dst)
    deg[3] = 0                                                   # an isolated destination aggregates to zero
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = rs.randint(0, n_src, int(deg.sum())).astype(np.int32)  # with repeats
    x = torch.tensor(rs.randn(n_src, c_in), dtype=torch.float32, requires_grad=True)
    w_l = torch.tensor(rs.randn(c_out, c_in), dtype=torch.float32, requires_grad=True)
    b_l = torch.tensor(rs.randn(c_out), dtype=torch.float32, requires_grad=True)
    w_r = torch.tensor(rs.randn(c_out, c_in), dtype=torch.float32, requires_grad=True)
    out = oracle.sage_conv_torch(x, torch.as_tensor(rowptr), torch.as_tensor(col), w_l, b_l, w_r)
    g = torch.tensor(rs.randn(n_dst, c_out), dtype=torch.float32)
    out.backward(g)
    A = np.zeros((n_dst, n_src))
    for i in range(n_dst):
        for p in range(rowptr[i], rowptr[i + 1]):
            A[i, col[p]] += 1.0
    Dinv = np.diag(1.0 / np.maximum(deg, 1))
    X, Wl, Bl, Wr = (t.detach().double().requires_grad_(True) for t in (x, w_l, b_l, w_r))
    M = torch.as_tensor(Dinv @ A)
    dense = M @ X @ Wl.T + Bl + X[:n_dst] @ Wr.T
    dense.backward(g.double())
    assert torch.allclose(out.detach().double(), dense.detach(), rtol=1e-5, atol=1e-5)
    for got, want in ((x.grad, X.grad), (w_l.grad, Wl.grad), (b_l.grad, Bl.grad), (w_r.grad, Wr.grad)):
        assert torch.allclose(got.double(), want, rtol=1e-4, atol=1e-5)
