"""Drop-in for the reference's ``utils`` module: ``from graphpope_amd.utils import Graphpope``.

Same names, argument meaning, return layout and error behaviour as /root/reference/utils.py for the
hot path; the arithmetic runs on the MI355X through libgraphpope_hip.so:

=============================  =============================================================
reference (utils.py)           here
=============================  =============================================================
sample_anchor_nodes :18-62     host NumPy ('stochastic' draws from the same global legacy RNG)
shortest_path_length :64-81    engine.bfs -- batched multi-source BFS kernel
all_pairs_..._parallel :92     (the mp.Pool fan-out is gone; num_workers is accepted and ignored)
get_geodesic_distance_vector   engine.build_csr + engine.bfs + engine.finalize
concat_into_features :129      host -> host: data.x copied on the host cores (pope_host_copy_2d), only the K
                               embedding columns cross PCIe; device-resident callers: finalise kernel / pope_concat
attach_distance_embedding      same prints, sets data.anchor_nodes
attach_node2vec :149           pairwise MFMA tile + column min-max kernel; K-means anchors: engine.kmeans_centers
Graphpope :182                 same signature, same process-lifetime cache
=============================  =============================================================

``data`` is duck-typed: ``.x`` float32 [N, F], ``.edge_index`` int64 [2, E], ``.num_nodes``.
The returned tensor is a NEW CPU float32 [N, F+K] tensor, as in the reference.
"""
from __future__ import annotations

import os
import os.path as osp

import numpy as np
import torch

from . import engine
from . import _lib as _lib_mod

# where attach_node2vec looks for {dataset}_node2vec.pt (reference: <dir of utils.py>/data, utils.py:155)
NODE2VEC_DIR = os.environ.get("GRAPHPOPE_DATA_DIR", osp.join(osp.dirname(osp.realpath(__file__)), "data"))

def _host_rankings():
    """utils.py:32-60: the one-off NetworkX rankings that stay on the host, call for call."""
    import networkx as nx
    return {
        "betweenness_centrality": nx.betweenness_centrality,                      # utils.py:34
        "eigenvector_centrality": nx.eigenvector_centrality_numpy,                # utils.py:46
        "clustering_coefficient": nx.clustering,                                  # utils.py:58
    }


_CENTRALITIES = ("betweenness_centrality", "eigenvector_centrality", "clustering_coefficient")


def _device():
    dev = engine.require_gpu()
    local_rank = os.environ.get("LOCAL_RANK")
    if local_rank is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
        dev = torch.device("cuda", int(local_rank) % torch.cuda.device_count())
        torch.cuda.set_device(dev)
    return dev


def _shard() -> bool:
    """Shard anchors over the ranks of an initialised process group unless GRAPHPOPE_SHARD=0."""
    return os.environ.get("GRAPHPOPE_SHARD", "1") != "0"


def sample_anchor_nodes(data, num_anchor_nodes, sampling_method):
    """utils.py:18-62.  'stochastic' is bit-identical (global legacy NumPy RNG, with replacement).

    'degree_centrality' is reproduced on the host from edge_index (in + out degree of the DiGraph
    to_networkx builds: repeated edges collapsed, ascending stable sort, last K kept: utils.py:38-42).
    'closeness_centrality' runs the multi-source BFS kernel from every node (engine.closeness_centrality) and
    reproduces NetworkX's scores bit for bit, hence the same anchors (utils.py:50-54).
    'pagerank' runs the SciPy power iteration of nx.pagerank as SpMV over the device CSR (engine.pagerank), scores
    bit-identical to NetworkX (utils.py:26-30).
    The remaining rankings (betweenness, eigenvector, clustering) are the reference's own one-off NetworkX calls, repeated
    on the host on the DiGraph to_networkx would build (anchor selection is not the accelerated
    path; the BFS from the chosen anchors is).
    """
    if sampling_method == "stochastic":
        node_indices = np.arange(data.num_nodes)
        return np.random.choice(node_indices, num_anchor_nodes)
    if sampling_method == "degree_centrality":
        ei = data.edge_index.detach().cpu().numpy().astype(np.int64)
        n = int(data.num_nodes)
        pairs = np.unique(ei[0] * n + ei[1])                       # DiGraph keeps one edge per (u, v)
        deg = np.bincount(pairs // n, minlength=n) + np.bincount(pairs % n, minlength=n)
        order = np.argsort(deg, kind="stable")                     # ascending, ties in node order
        return order[-num_anchor_nodes:].tolist()
    if sampling_method == "closeness_centrality":
        # utils.py:50-54.  All-sources BFS on the GPU (exact integers), NetworkX's float formula on the host: same scores,
        # same ascending stable sort, same last-K keys.
        ei = data.edge_index.detach().to(_device(), torch.int64)
        score = engine.closeness_centrality(ei, int(data.num_nodes))
        order = np.argsort(score, kind="stable")
        return order[-num_anchor_nodes:].tolist()
    if sampling_method == "pagerank":
        # utils.py:26-30 nx.pagerank_scipy (folded into nx.pagerank in NetworkX 3: the same SciPy power iteration), as SpMV
        # iterations over the device CSR; float64 scores bit-identical to NetworkX, hence the same last-K keys.
        ei = engine.stage_to_device(data.edge_index.detach(), _device()).to(torch.int64)
        score = engine.pagerank(ei, int(data.num_nodes))
        order = np.argsort(score, kind="stable")
        return order[-num_anchor_nodes:].tolist()
    if sampling_method in _CENTRALITIES:
        import networkx as nx
        ei = data.edge_index.detach().cpu().numpy()
        G = nx.DiGraph()                                           # torch_geometric.utils.to_networkx(data), utils.py:27
        G.add_nodes_from(range(int(data.num_nodes)))
        G.add_edges_from(zip(ei[0].tolist(), ei[1].tolist()))
        score = _host_rankings()[sampling_method](G)
        ranked = {k: v for k, v in sorted(score.items(), key=lambda item: item[1])}      # ascending, ties in node order
        return list(ranked.keys())[-num_anchor_nodes:]
    # the reference falls through every `if` and hits `return sampled_anchor_nodes` unbound (utils.py:62)
    raise UnboundLocalError("local variable 'sampled_anchor_nodes' referenced before assignment")


def _host_features(data):
    x = data.x.detach()
    if not (x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        x = x.to(torch.float32).contiguous()
    return x


def _assemble_on_host(data, k, embedding_fn):
    """The tensor the call returns: a NEW contiguous CPU float32 [N, F+K] (utils.py:134) in ordinary pageable memory, like
    the reference's.  (Rounds 1-2 returned a pinned tensor: 34 ms of hipHostMalloc on the one call a process makes, and
    270 MB -- 8.6 GB for R-MAT-22 x 512 -- page-locked for as long as the memoised result lives.)

    out[:, :F] = data.x is copied host to host by a few threads (data.x never crosses PCIe); out[:, F:] = the device
    embedding (utils.py:129-135 torch.cat((data.x, embedding), 1)) comes over in 8 MB chunks through a ring of three pinned
    slots allocated once per process, and the same threads copy each landed chunk out while the next is on the bus; without
    the ring (allocation refused) the columns are staged through the library's 4 MB bounce buffer.  No page of the result is
    ever handed to the HIP runtime.  ``GRAPHPOPE_HOST_RESULT=staged`` additionally keeps the caller's edge_index pages away from
    hipHostRegister (the upload goes through pinned staging memory).  (Rounds 3-4 also had a ``registered`` mode -- the result's
    own pages registered chunk by chunk, 11.4 ms first call and 4-18 ms on repeated calls -- and a ``pinned`` one; both were
    slower and were removed in round 5.)  On a FIRST call (fresh result pages) the assembly starts AFTER ``embedding_fn()`` has
    uploaded edge_index and run the GPU work: sixteen threads faulting in a quarter gigabyte of huge pages beside it stalled the
    GPU queues for milliseconds (3.4-3.9 ms in a fresh process, 13-20 ms inside bench.py).  On a repeated call the pages come
    from the pool, nothing faults, and the feature copy runs underneath the upload and the GPU work."""
    import time as _t
    trace = os.environ.get("GRAPHPOPE_TRACE")
    t0 = _t.perf_counter()
    x = _host_features(data)
    n, f = int(x.shape[0]), int(x.shape[1])
    mode = os.environ.get("GRAPHPOPE_HOST_RESULT", "ring")
    if mode not in ("ring", "staged"):
        raise ValueError(f"GRAPHPOPE_HOST_RESULT={mode!r}: expected ring or staged")
    _lib_mod.load().pope_assemble_prepare(torch.cuda.current_device())     # first call of a process: the pinned ring is allocated beside the GPU work
    out, reused = engine.host_result_tensor(n, f + k, with_origin=True)
    asm = None
    try:
        # pages from the pool (a repeated call): no page faults to take, so the feature copy may start now and run underneath
        # the upload and the GPU work; fresh pages (the first call): the copy starts after the GPU work
        if reused:
            asm = engine.HostAssembly(x if f else None, out, f)
        emb_dev = embedding_fn()                   # float32 [N, K], or (uint8 codes [N, K], float32 lut [256]): engine.hop_codes
        coded = isinstance(emb_dev, tuple)
        if not coded:
            emb_dev = emb_dev.contiguous()
        t1 = _t.perf_counter()
        if asm is None:
            asm = engine.HostAssembly(x if f else None, out, f)
        with asm:
            res = asm.finish_codes(*emb_dev) if coded else asm.finish(emb_dev)
    except BaseException:
        if asm is not None:
            asm.__exit__(None, None, None)         # waits for the host threads of an assembly that will not be finished
        raise
    if trace:
        import sys as _s
        print(f"[trace] upload + GPU {1e3 * (t1 - t0):.2f} ms, result assembly {1e3 * (_t.perf_counter() - t1):.2f} ms", file=_s.stderr)
    return res


def _geodesic_planes(ei, n, anchors, dev):
    """Hop planes of all anchors on this rank's GPU, through the persisted cache if GRAPHPOPE_CACHE_DIR is set
    (SURVEY.md §8f rank 4): the reference only memoises inside one process (utils.py:195-208); here the bit-sliced hop
    planes survive on disk, keyed by the graph and the anchor list, so re-runs with the same seed (or other ranks /
    later processes) skip the BFS and only expand the planes."""
    from . import plane_cache
    cache_dir = os.environ["GRAPHPOPE_CACHE_DIR"]
    key = plane_cache.graph_key(ei, n, anchors)
    hit = plane_cache.load(cache_dir, key, dev)
    if hit is not None:
        return hit
    _, hp = engine.geodesic_run(None, ei, n, anchors, want_out=False)
    plane_cache.save(cache_dir, key, hp, anchors)
    return hp.valid(), hp.n_hop_bits


def _geodesic_embedding_device(edge_index, n, anchors, dev, coded=False):
    """float32 [N, K] on the device: 1 / (hops + 1) to every anchor (sharded over the ranks of a process group).

    ``coded=True`` (the host -> host caller, one rank): the transport form instead when the hop counts allow it --
    (uint8 codes [N, K], float32 lut [256]) from engine.hop_codes, a quarter of the bytes to bring down."""
    from . import distributed as pdist
    trace = os.environ.get("GRAPHPOPE_TRACE")
    import time as _t
    t0 = _t.perf_counter()
    register = os.environ.get("GRAPHPOPE_HOST_RESULT", "ring") != "staged"
    with engine.staged(edge_index.detach(), dev, register=register) as ei_dev:   # 14.4 MB straight from the caller's pages, released on exit
        ei = ei_dev.to(torch.int64)
        if not os.environ.get("GRAPHPOPE_CACHE_DIR"):
            t1 = _t.perf_counter()
            if coded and (not _shard() or pdist.world_size(None) == 1):
                _, hp = engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)
                # the byte transport needs the process's pinned ring (allocated beside the BFS above: pope_assemble_prepare)
                if hp.max_hop <= engine.MAX_CODED_HOP and _lib_mod.load().pope_assemble_ring_ready():
                    emb = engine.hop_codes(hp)
                else:                                                # a graph deeper than a byte: the float columns
                    emb = torch.empty((n, len(anchors)), dtype=torch.float32, device=dev)
                    engine.finalize(hp.valid().contiguous(), hp.n_hop_bits, n, len(anchors), None, 0, emb, 0)
            else:
                emb = engine.geodesic_features(None, ei, n, anchors, shard=_shard())
            if trace:
                import sys as _s
                print(f"[trace] stage {1e3 * (t1 - t0):.2f} ms, geodesic_features {1e3 * (_t.perf_counter() - t1):.2f} ms", file=_s.stderr)
            return emb
        planes, bits = _geodesic_planes(ei, n, anchors, dev)
    emb = torch.empty((n, len(anchors)), dtype=torch.float32, device=dev)
    engine.finalize(planes.contiguous(), bits, n, len(anchors), None, 0, emb, 0)
    return emb


def _geodesic_features(data, dev):
    """[N, F+K] for the geodesic branch.  Host-resident data.x (the reference's case): only edge_index goes up and only
    the K embedding columns come down.  Device-resident data.x: everything stays on the device."""
    n, anchors = int(data.num_nodes), np.asarray(data.anchor_nodes, dtype=np.int64)
    if data.x.is_cuda:
        x = data.x.detach().to(dev, torch.float32)
        emb_cols = _geodesic_embedding_device(data.edge_index, n, anchors, dev)
        out = torch.empty((n, x.shape[1] + len(anchors)), dtype=torch.float32, device=dev)
        out[:, : x.shape[1]] = x
        out[:, x.shape[1]:] = emb_cols
        return out.cpu()
    transport = os.environ.get("GRAPHPOPE_HOST_TRANSPORT", "codes")
    if transport not in ("codes", "float"):
        raise ValueError(f"GRAPHPOPE_HOST_TRANSPORT={transport!r}: expected codes or float")
    coded = transport == "codes" and os.environ.get("GRAPHPOPE_HOST_RESULT", "ring") == "ring"
    return _assemble_on_host(data, len(anchors), lambda: _geodesic_embedding_device(data.edge_index, n, anchors, dev, coded))


def get_geodesic_distance_vector(data, num_workers):
    """utils.py:116-126: float32 [N, K], entry (v, j) = 1 / (hops(v -> anchor_nodes[j]) + 1), 0 if no path.

    ``num_workers`` (CPU processes in the reference) is accepted for signature parity and unused.
    """
    dev = _device()
    n = int(data.num_nodes)
    csr = engine.build_csr(engine.stage_to_device(data.edge_index.detach(), dev).to(torch.int64), n)
    hp = engine.bfs(csr, data.anchor_nodes)
    out = torch.empty((n, hp.k), dtype=torch.float32, device=dev)
    engine.finalize(hp.planes, hp.n_hop_bits, n, hp.k, None, 0, out, 0)
    return out.cpu()


def concat_into_features(embedding_matrix, data):
    """utils.py:129-135."""
    embedding_tensor = torch.as_tensor(embedding_matrix)
    return torch.cat((data.x, embedding_tensor), 1)


def attach_distance_embedding(data, dataset, num_anchor_nodes, sampling_method, distance_function, num_workers):
    """utils.py:137-147 (ignores dataset / distance_function, like the reference)."""
    print('sampling anchor nodes...')
    data.anchor_nodes = sample_anchor_nodes(data=data, num_anchor_nodes=num_anchor_nodes, sampling_method=sampling_method)
    print('deriving shortest paths to anchor nodes...')
    if len(data.anchor_nodes) == 0:                                  # --num_anchor_nodes 0: cat((x, [N, 0])) is a copy of x
        print('feature matrix is blessed by the POPE!')
        return data.x.detach().clone()
    extended_features = _geodesic_features(data, _device())
    print('feature matrix is blessed by the POPE!')
    return extended_features


def attach_node2vec(data, dataset, num_anchor_nodes, sampling_method, distance_function, num_workers):
    """utils.py:149-180.  KeyError for an unknown distance_function, FileNotFoundError for a missing table."""
    print('sampling anchor nodes...')
    loading_path = osp.join(NODE2VEC_DIR, f'{dataset}_node2vec.pt')
    node2vec_embeddings = torch.load(loading_path, map_location="cpu").detach()
    if distance_function not in ('distance', 'similarity', 'euclidean'):
        raise KeyError(distance_function)
    anchor_nodes = anchor_embeddings = None
    dev = _device()
    table = engine.stage_to_device(node2vec_embeddings.to(torch.float32), dev)
    if sampling_method == 'stochastic':
        anchor_nodes = sample_anchor_nodes(data, num_anchor_nodes, sampling_method='stochastic')
    else:
        # utils.py:168-170: every other sampling_method means K-means centres as anchors.  Anchor SELECTION is not the
        # accelerated path (the N x K distance matrix is), and K-means is discontinuous in its inputs: the default is
        # the reference's own call, which reproduces its centres exactly from the same global NumPy stream.
        # GRAPHPOPE_KMEANS=gpu opts into engine.kmeans_centers -- the same algorithm (k-means++ from the same stream,
        # Lloyd, scikit-learn's empty-cluster rule) with the arithmetic on the GPU: 0.17 s instead of minutes at Flickr
        # size, centres equal to scikit-learn's on separated data and equal IN DISTRIBUTION on overlapping data.
        if os.environ.get("GRAPHPOPE_KMEANS", "sklearn").lower() == "gpu":
            anchor_embeddings = engine.kmeans_centers(table, num_anchor_nodes)
        else:
            from sklearn.cluster import KMeans
            anchor_embeddings = KMeans(n_clusters=num_anchor_nodes).fit(node2vec_embeddings.numpy()).cluster_centers_
        print('K means cluster anchor nodes derived!')
    if data.x.is_cuda:
        extended_features = engine.pairwise_features(data.x.detach().to(dev, torch.float32), table, anchor_nodes,
                                                     distance_function, anchor_embeddings=anchor_embeddings).cpu()
    else:
        k = len(anchor_embeddings) if anchor_embeddings is not None else len(anchor_nodes)
        extended_features = _assemble_on_host(data, k, lambda: engine.pairwise_embedding(table, anchor_nodes, distance_function,
                                                                                         anchor_embeddings=anchor_embeddings))
    print('feature matrix is blessed by the POPE')
    return extended_features


_cached_pope_embedding = None


def clear_cache():
    """Forget the memoised result (the reference can only do this by restarting the process)."""
    global _cached_pope_embedding
    _cached_pope_embedding = None


def Graphpope(data, dataset: str, embedding_space: str, sampling_method: str, num_anchor_nodes: int,
              distance_function=None, num_workers=4):
    """utils.py:182-210: returns the feature matrix with the GraphPOPE embedding in its last K columns.

    The first result is memoised for the life of the process and returned -- the same object,
    whatever the arguments -- by every later call (utils.py:201-208: trainer.test() re-runs setup()).
    Unknown embedding_space -> KeyError.
    """
    global _cached_pope_embedding
    pope_map = {
        'geodesic': attach_distance_embedding,
        'node2vec': attach_node2vec,
    }
    if _cached_pope_embedding is None:
        pope = pope_map[embedding_space]
        _cached_pope_embedding = pope(data, dataset, num_anchor_nodes, sampling_method, distance_function,
                                      num_workers=num_workers)
    return _cached_pope_embedding
