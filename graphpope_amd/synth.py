"""Synthetic graphs of the shapes SURVEY.md §8(d) names (no dataset files exist offline).

NumPy only, deterministic in `seed`.  Every generator returns ``edge_index`` as an
``int64 [2, E]`` array in the PyG convention the reference consumes
(/root/reference/utils.py:121 ``to_networkx(data)``: column ``e`` is the directed edge
``edge_index[0, e] -> edge_index[1, e]``).
"""
from __future__ import annotations

import numpy as np

FLICKR_N = 89_250          # /root/reference/main.py:77-83 (Flickr: 500 features, 7 classes)
FLICKR_PAIRS = 449_878     # undirected pairs -> E = 899 756 directed entries
PUBMED_N = 19_717
PUBMED_PAIRS = 44_324


def _symmetrise(src: np.ndarray, dst: np.ndarray, n: int) -> np.ndarray:
    """Undirected pairs -> both directions, self-loops and duplicates removed, sorted by (row, col)."""
    keep = src != dst
    src, dst = src[keep], dst[keep]
    lo = np.minimum(src, dst).astype(np.int64)
    hi = np.maximum(src, dst).astype(np.int64)
    key = np.unique(lo * n + hi)
    lo, hi = key // n, key % n
    row = np.concatenate([lo, hi])
    col = np.concatenate([hi, lo])
    order = np.lexsort((col, row))
    return np.stack([row[order], col[order]]).astype(np.int64)


def powerlaw_graph(n: int, pairs: int, seed: int = 1, alpha: float = 0.62, shift: float = 3.0) -> np.ndarray:
    """Chung-Lu style graph with exactly `pairs` distinct undirected edges.

    Endpoint i is drawn with probability proportional to (i + shift) ** -alpha, which gives the
    long-tailed degree sequence (mean ~2*pairs/n, a few hubs with thousands of neighbours) of the
    Flickr / PubMed graphs; vertex ids are permuted so degree is not monotone in the id.
    """
    rng = np.random.default_rng(seed)
    w = (np.arange(n, dtype=np.float64) + shift) ** (-alpha)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    keys = np.empty(0, dtype=np.int64)
    while keys.size < pairs:
        need = int((pairs - keys.size) * 1.3) + 1024
        a = np.searchsorted(cdf, rng.random(need))
        b = np.searchsorted(cdf, rng.random(need))
        ok = a != b
        lo = np.minimum(a[ok], b[ok]).astype(np.int64)
        hi = np.maximum(a[ok], b[ok]).astype(np.int64)
        new = lo * n + hi
        # keep first-seen order so the result does not depend on the batch size
        allk = np.concatenate([keys, new])
        _, first = np.unique(allk, return_index=True)
        keys = allk[np.sort(first)]
    keys = keys[:pairs]
    perm = rng.permutation(n)
    return _symmetrise(perm[keys // n], perm[keys % n], n)


def flickr_like(seed: int = 1) -> tuple[np.ndarray, int]:
    """Config 2/3/4 graph: N = 89 250, E = 899 756 directed entries (symmetric)."""
    return powerlaw_graph(FLICKR_N, FLICKR_PAIRS, seed=seed, alpha=0.71, shift=1.7), FLICKR_N


def pubmed_like(seed: int = 1) -> tuple[np.ndarray, int]:
    """Config 1 graph: N = 19 717, E = 88 648 directed entries (symmetric), max degree ~170."""
    return powerlaw_graph(PUBMED_N, PUBMED_PAIRS, seed=seed, alpha=0.6, shift=10.0), PUBMED_N


def rmat(scale: int, edge_factor: int = 8, seed: int = 1, abcd=(0.57, 0.19, 0.19, 0.05),
         symmetric: bool = True, permute: bool = True) -> tuple[np.ndarray, int]:
    """Config 5 generator: R-MAT, `edge_factor` undirected edges per node before dedupe."""
    n = 1 << scale
    m = n * edge_factor
    rng = np.random.default_rng(seed)
    a, b, c, _ = abcd
    src = np.zeros(m, dtype=np.int64)
    dst = np.zeros(m, dtype=np.int64)
    for _bit in range(scale):
        r = rng.random(m)
        right = (r >= a) & (r < a + b) | (r >= a + b + c)      # quadrants b, d -> dst bit set
        down = r >= a + b                                        # quadrants c, d -> src bit set
        src = (src << 1) | down
        dst = (dst << 1) | right
    if permute:
        perm = rng.permutation(n)
        src, dst = perm[src], perm[dst]
    if symmetric:
        return _symmetrise(src, dst, n), n
    keep = src != dst
    key = np.unique(src[keep] * n + dst[keep])
    return np.stack([key // n, key % n]).astype(np.int64), n


def seeded_anchors(n: int, k: int, seed: int = 42) -> np.ndarray:
    """The reference's stochastic draw (/root/reference/utils.py:22-24) on a fresh legacy RNG.

    ``np.random.seed(seed); np.random.choice(np.arange(n), k)`` -- with replacement, draw order kept.
    RandomState(seed) reproduces the global legacy stream without touching global state.
    """
    return np.random.RandomState(seed).choice(np.arange(n), k)
