"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU restatement of the GraphPOPE hot path, used only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  ``graphpope_amd`` never imports it.

Parity status
-------------
* geodesic (hops, f32 embedding, concat): **pinned** -- checked bit-for-bit against
  ``tests/golden/geodesic_*.npz``, which are outputs of the reference's own ``utils.py`` run in the
  build container by ``tests/golden/make_goldens.py`` (the reference has no tests or fixtures of
  its own).
* node2vec pairwise + min-max: **pinned to the container's scikit-learn 1.7.2** through
  ``tests/golden/node2vec_*.npz`` (same script).  The reference pins scikit-learn 0.24.2
  (requirements.txt:4), which is not installable offline, so float parity against *that* version
  is unpinned; tolerance 1e-5 abs.
* SAGEConv: **parity unpinned** -- PyG/torch_sparse are absent, the torch fp32 restatement below
  is the only checker (op-level, self-referential; SURVEY.md §8c last row).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    """Compile oracle/pope_oracle.c with gcc (building the checker is not using it)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return os.path.join(_HERE, "libpope_oracle.so")


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpope_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(
                os.path.join(_HERE, "pope_oracle.c")):
            build()
        lib = ctypes.CDLL(path)
        lib.oracle_geodesic_hops.restype = ctypes.c_int
        lib.oracle_geodesic_hops.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        lib.oracle_hops_to_embedding.restype = None
        lib.oracle_hops_to_embedding.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        lib.oracle_concat.restype = None
        lib.oracle_concat.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _LIB = lib
    return _LIB


# --------------------------------------------------------------------------------------------
# geodesic  (/root/reference/utils.py:64-135)
# --------------------------------------------------------------------------------------------
def geodesic_hops(edge_index: np.ndarray, num_nodes: int, anchors) -> np.ndarray:
    """int32 [N, K]; hops[v, j] = edges on a shortest path v -> anchors[j], -1 if unreachable.

    Follows utils.py:64-81 (value definition) and :92-107 (row i = node i).  One BFS per anchor over
    reversed edges instead of one bidirectional BFS per pair -- same integers.
    """
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    assert ei.ndim == 2 and ei.shape[0] == 2
    anc = np.ascontiguousarray(np.asarray(anchors), dtype=np.int64)
    hops = np.empty((num_nodes, anc.size), dtype=np.int32)
    rc = _lib().oracle_geodesic_hops(ei.ctypes.data, ei.shape[1], num_nodes, anc.ctypes.data,
                                     anc.size, hops.ctypes.data)
    if rc == -2:
        raise IndexError("edge_index / anchor id outside [0, num_nodes)")
    if rc:
        raise MemoryError("oracle_geodesic_hops")
    return hops


def hops_to_embedding(hops: np.ndarray) -> np.ndarray:
    """float32, 1/(hops+1), unreachable -> 0  (utils.py:73,75-76,125)."""
    h = np.ascontiguousarray(hops, dtype=np.int32)
    emb = np.empty(h.shape, dtype=np.float32)
    _lib().oracle_hops_to_embedding(h.ctypes.data, h.size, emb.ctypes.data)
    return emb


def concat_into_features(x: np.ndarray, emb: np.ndarray) -> np.ndarray:
    """utils.py:129-135."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    emb = np.ascontiguousarray(emb, dtype=np.float32)
    out = np.empty((x.shape[0], x.shape[1] + emb.shape[1]), dtype=np.float32)
    _lib().oracle_concat(x.ctypes.data, x.shape[0], x.shape[1], emb.ctypes.data, emb.shape[1],
                         out.ctypes.data)
    return out


def geodesic_features(x, edge_index, num_nodes, anchors) -> np.ndarray:
    """utils.py:137-147 given the anchors: [N, F+K] float32."""
    return concat_into_features(x, hops_to_embedding(geodesic_hops(edge_index, num_nodes, anchors)))


def sample_anchor_nodes_stochastic(num_nodes: int, num_anchor_nodes: int) -> np.ndarray:
    """utils.py:22-24 -- draws from the GLOBAL legacy NumPy RNG, with replacement."""
    return np.random.choice(np.arange(num_nodes), num_anchor_nodes)


def geodesic_pairs_networkx(edge_index: np.ndarray, num_nodes: int, anchors, nodes) -> np.ndarray:
    """The reference's actual CPU algorithm, statement for statement (utils.py:64-81, 116-121):
    a NetworkX DiGraph and one ``nx.shortest_path`` (bidirectional BFS) per (node, anchor) pair.
    Used for small cross-checks and as bench.py's reference-equivalent CPU baseline.  float32 [len(nodes), K].
    """
    import networkx as nx
    G = nx.DiGraph()
    G.add_nodes_from(range(num_nodes))
    G.add_edges_from(zip(edge_index[0].tolist(), edge_index[1].tolist()))
    rows = []
    for node in nodes:
        row = []
        for a in anchors:
            try:
                row.append(1 / len(nx.shortest_path(G, source=int(node), target=int(a))))
            except nx.NetworkXNoPath:
                row.append(0)
        rows.append(row)
    return np.asarray(rows, dtype=np.float32).reshape(len(rows), len(anchors))


def _pairs_job(G, anchors, partition):
    """utils.py:64-81 shortest_path_length: {node: [1 / len(path) or 0 per anchor]} for one partition of the nodes."""
    import networkx as nx
    dists_dict = {}
    for node in partition:
        distances = []
        for anchor_node in anchors:
            try:
                distances.append(1 / len(nx.shortest_path(G, source=node, target=anchor_node)))
            except nx.NetworkXNoPath:
                distances.append(0)
        dists_dict[node] = distances.copy()
    return dists_dict


def geodesic_pairs_networkx_pool(edge_index: np.ndarray, num_nodes: int, anchors, nodes, workers: int):
    """The reference's CPU path as it actually runs (utils.py:92-107, 116-121): NetworkX DiGraph, the node list cut into
    `workers` float-indexed slices, one ``multiprocessing.Pool(workers).apply_async`` job per slice (the whole graph is
    pickled to every job), results merged in slice order.  `nodes` is the (sub)list of nodes to process -- bench.py's
    bounded sample; the reference passes all of them.  Returns (float32 [len(nodes), K], {"graph_build_s", "pool_s"}).
    """
    import multiprocessing as mp
    import time
    import networkx as nx
    t0 = time.perf_counter()
    G = nx.DiGraph()
    G.add_nodes_from(range(num_nodes))
    G.add_edges_from(zip(edge_index[0].tolist(), edge_index[1].tolist()))
    t1 = time.perf_counter()
    nodes = [int(v) for v in nodes]
    anchors = [int(a) for a in anchors]
    n = len(nodes)
    pool = mp.Pool(workers)
    try:
        jobs = [pool.apply_async(_pairs_job, (G, anchors, nodes[int(n / workers * i):int(n / workers * (i + 1))]))
                for i in range(workers)]
        merged = {}
        for job in jobs:
            merged.update(job.get())
    finally:
        pool.close()
        pool.join()
    t2 = time.perf_counter()
    rows = [merged[v] for v in nodes if v in merged]              # utils.py:100 can drop the last node for some (n, workers)
    return (np.asarray(rows, dtype=np.float32).reshape(len(rows), len(anchors)),
            {"graph_build_s": t1 - t0, "pool_s": t2 - t1})


def pagerank_scores(edge_index: np.ndarray, num_nodes: int, alpha=0.85, max_iter=100, tol=1.0e-6) -> np.ndarray:
    """nx.pagerank_scipy(to_networkx(data)) (utils.py:26-30; NetworkX 3: nx.pagerank -> _pagerank_scipy) restated without
    NetworkX / SciPy: DiGraph semantics (one edge per distinct pair), x @ A accumulated per target over its sources in
    ascending order (SciPy's csc_matvec on A^T), Python sum for the dangling mass, NumPy sum for the l1 norm.  Bit-identical
    to NetworkX 3.4.2 + SciPy 1.15.3 (tests/test_oracle.py).  Pure-Python inner loop: small graphs only."""
    n = int(num_nodes)
    ei = np.asarray(edge_index, dtype=np.int64)
    key = np.unique(ei[0] * n + ei[1])
    src, dst = key // n, key % n
    outdeg = np.bincount(src, minlength=n).astype(np.float64)
    w = np.zeros(n)
    nz = outdeg != 0
    w[nz] = 1.0 / outdeg[nz]
    dangling = np.where(outdeg == 0)[0]
    order = np.lexsort((src, dst))
    rsrc = src[order].tolist()
    rptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))]).tolist()
    x = np.repeat(1.0 / n, n)
    p = np.repeat(1.0 / n, n)
    for _ in range(max_iter):
        xlast = x
        xl, wl = x.tolist(), w.tolist()
        y = np.zeros(n)
        for i in range(n):
            acc = 0.0
            for q in range(rptr[i], rptr[i + 1]):
                j = rsrc[q]
                acc += wl[j] * xl[j]
            y[i] = acc
        x = alpha * (y + sum(x[dangling]) * p) + (1 - alpha) * p
        if np.absolute(x - xlast).sum() < n * tol:
            return x
    raise RuntimeError("pagerank: power iteration failed to converge")


# --------------------------------------------------------------------------------------------
# node2vec-space pairwise + min-max  (/root/reference/utils.py:149-180)
# --------------------------------------------------------------------------------------------
def _normalize_rows(a: np.ndarray) -> np.ndarray:
    # sklearn.preprocessing.normalize (l2): x / ||x||, zero rows left as zero; dtype kept (f32).
    norms = np.sqrt(np.einsum("ij,ij->i", a, a))
    norms[norms == 0.0] = 1.0
    return a / norms[:, None]


def pairwise(x: np.ndarray, anchors_emb: np.ndarray, distance_function: str) -> np.ndarray:
    """utils.py:158-174 restated after scikit-learn's algorithms.

    'similarity' -> cosine_similarity  (pairwise.py:1683-1738: normalise rows, f32 GEMM)
    'distance'   -> cosine_distances   (pairwise.py:1129-1175: 1 - S clipped to [0, 2])
    'euclidean'  -> euclidean_distances (pairwise.py:391-442, 582-653: f32 inputs are upcast to f64,
                    d2 = xx + yy - 2 x.y in f64, cast to f32, clamp at 0, sqrt in f32)
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    a = np.ascontiguousarray(anchors_emb, dtype=np.float32)
    if distance_function in ("similarity", "distance"):
        s = _normalize_rows(x) @ _normalize_rows(a).T
        if distance_function == "similarity":
            return s.astype(np.float32)
        return np.clip(np.float32(1.0) - s, 0.0, 2.0).astype(np.float32)
    if distance_function == "euclidean":
        x64, a64 = x.astype(np.float64), a.astype(np.float64)
        d2 = -2.0 * (x64 @ a64.T)
        d2 += np.einsum("ij,ij->i", x64, x64)[:, None]
        d2 += np.einsum("ij,ij->i", a64, a64)[None, :]
        d = d2.astype(np.float32)
        np.maximum(d, 0, out=d)
        return np.sqrt(d, out=d)
    raise KeyError(distance_function)           # utils.py:164: dist_map[distance_function]


def minmax_scale_columns(e: np.ndarray) -> np.ndarray:
    """MinMaxScaler().fit(E).transform(E)  (utils.py:175-176; sklearn _data.py:92-124, 456-567).

    Per COLUMN over all rows; scale_ = 1/range with range < 10*eps treated as 1; X*scale_ + min_.
    """
    e = np.asarray(e, dtype=np.float32)
    dmin = e.min(axis=0)
    dmax = e.max(axis=0)
    rng = dmax - dmin
    rng = np.where(rng < 10 * np.finfo(np.float32).eps, np.float32(1.0), rng).astype(np.float32)
    scale = (np.float32(1.0) / rng).astype(np.float32)
    mn = (np.float32(0.0) - dmin * scale).astype(np.float32)
    return (e * scale + mn).astype(np.float32)


def node2vec_features(x, node2vec_emb, anchors, distance_function, anchor_embeddings=None) -> np.ndarray:
    """utils.py:149-180 given the anchors: [N, F+K] float32.  Stochastic branch: `anchors` are row indices
    (utils.py:165-167); K-means branch: `anchor_embeddings` are the cluster centres (utils.py:168-170)."""
    emb = np.asarray(node2vec_emb, dtype=np.float32)
    a = emb[np.asarray(anchors, dtype=np.int64)] if anchor_embeddings is None else np.asarray(anchor_embeddings)
    return concat_into_features(x, minmax_scale_columns(pairwise(emb, a, distance_function)))


# --------------------------------------------------------------------------------------------
# SAGEConv  (/root/reference/main.py:204-211 + PyG 1.7.0 SAGEConv, absent here: believed semantics)
# --------------------------------------------------------------------------------------------
def sage_conv_torch(x_src, rowptr, col, w_l, b_l, w_r):
    """out[i] = lin_l(mean_{j in N(i)} x_src[j]) + lin_r(x_src[i]),  i < n_dst = len(rowptr)-1.

    Plain torch fp32, differentiable.  Rows with no neighbours aggregate to zero (torch_sparse
    ``matmul(adj_t, x, reduce='mean')`` semantics).  lin_l has a bias, lin_r has none.
    """
    import torch
    n_dst = rowptr.numel() - 1
    deg = (rowptr[1:] - rowptr[:-1]).to(torch.int64)
    seg = torch.repeat_interleave(torch.arange(n_dst, device=x_src.device), deg)
    agg = torch.zeros(n_dst, x_src.shape[1], dtype=x_src.dtype, device=x_src.device)
    agg.index_add_(0, seg, x_src.index_select(0, col.to(torch.int64)))
    agg = agg / deg.clamp(min=1).to(x_src.dtype)[:, None]
    out = torch.nn.functional.linear(agg, w_l, b_l)
    return out + torch.nn.functional.linear(x_src[:n_dst], w_r)


# --------------------------------------------------------------------------------------------
# Fan-out sampling (stands in for PyG NeighborSampler, main.py:100-116): CPU restatement of
# graphpope_amd/csrc/sampler.hip -- same keyed Feistel permutation, same first-appearance relabelling.
# --------------------------------------------------------------------------------------------
_M32 = 0xFFFFFFFF


def _mix32(h):
    h &= _M32
    h ^= h >> 15
    h = (h * 0x2C1B3C6D) & _M32
    h ^= h >> 12
    h = (h * 0x297A2D39) & _M32
    h ^= h >> 15
    return h


def _row_key(seed, hop, node):
    lo, hi = seed & _M32, (seed >> 32) & _M32
    return _mix32(lo ^ _mix32((hi + 0x9E3779B1 * (hop + 1)) & _M32) ^ _mix32((node * 0x85EBCA6B + 0x165667B1) & _M32))


def feistel_perm(i, d, key):
    bits = 2
    while (1 << bits) < d:
        bits += 1
    hb = (bits + 1) >> 1
    mask = (1 << hb) - 1
    x = i
    while True:
        l, r = x >> hb, x & mask
        for rnd in range(4):
            f = _mix32((r * 0x9E3779B1 + key + rnd * 0x85EBCA6B) & _M32) & mask
            l, r = r, l ^ f
        x = (l << hb) | r
        if x < d:
            return x


def sample_hop(rowptr, col, targets, fanout, seed, hop):
    """One hop: (out_rowptr, out_col local ids, n_id) exactly as sage_sample_hop produces them."""
    targets = [int(t) for t in targets]
    local = {g: i for i, g in enumerate(targets)}
    n_id = list(targets)
    out_rowptr, out_col = [0], []
    for g in targets:
        beg, d = int(rowptr[g]), int(rowptr[g + 1] - rowptr[g])
        c = d if (fanout < 0 or d <= fanout) else fanout
        key = _row_key(seed, hop, g)
        for j in range(c):
            u = int(col[beg + (j if d <= c else feistel_perm(j, d, key))])
            if u not in local:
                local[u] = len(n_id)
                n_id.append(u)
            out_col.append(local[u])
        out_rowptr.append(len(out_col))
    return np.asarray(out_rowptr, np.int32), np.asarray(out_col, np.int32), np.asarray(n_id, np.int64)
