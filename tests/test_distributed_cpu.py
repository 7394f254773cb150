"""The N > 1 path on CPU: anchor sharding -> all-gather of hop planes -> reassembly, world_size 2 and 3 over gloo.

The BFS and the plane expansion are injected (NumPy + the oracle stand in for the HIP kernels, writing the
documented plane format of include/graphpope_hip.h), so this covers exactly the host logic of
graphpope_amd/distributed.py that the RCCL run uses on the GPUs.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def words_for(k):
    w = (k + 63) // 64
    return max(w, 1) if w <= 2 else (w + 3) // 4 * 4


class FakePlanes:
    def __init__(self, planes, bits):
        self.planes, self.n_hop_bits = planes, bits


def encode_planes(hops, capacity=8):
    """int32 [N, K] (-1 unreachable) -> int64 [capacity + 1, N, W] in the library's bit-sliced layout."""
    n, k = hops.shape
    w = words_for(k)
    planes = np.zeros((capacity + 1, n, w), dtype=np.uint64)
    max_hop = max(int(hops.max()), 0)
    bits = max_hop.bit_length()
    for j in range(k):
        word, bit = j // 64, np.uint64(j % 64)
        reach = hops[:, j] >= 0
        planes[0, reach, word] |= np.uint64(1) << bit
        for b in range(bits):
            sel = reach & (((hops[:, j] >> b) & 1) == 1)
            planes[1 + b, sel, word] |= np.uint64(1) << bit
    planes[1 + bits:] = np.uint64(0xDEADBEEFDEADBEEF)          # "need not be initialised": must never be read
    return torch.from_numpy(planes.view(np.int64)), bits


def numpy_finalize(planes, bits, n, k, x, f, out, c0):
    p = planes.numpy().view(np.uint64)
    emb = np.zeros((n, k), dtype=np.float32)
    for j in range(k):
        word, bit = j // 64, np.uint64(j % 64)
        reach = ((p[0, :, word] >> bit) & np.uint64(1)).astype(bool)
        h = np.zeros(n, dtype=np.int64)
        for b in range(bits):
            h |= ((p[1 + b, :, word] >> bit) & np.uint64(1)).astype(np.int64) << b
        emb[:, j] = np.where(reach, (1.0 / (h + 1)).astype(np.float32), np.float32(0))
    if x is not None:
        out[:, :f] = x
    out[:, f + c0: f + c0 + k] = torch.from_numpy(emb)


class FakePending:
    """What engine.PendingBfs offers, from the oracle: seen + 4 hop-bit planes now, the verdict at finish()."""

    def __init__(self, hops):
        planes, bits = encode_planes(hops)
        self._hp = FakePlanes(planes, bits)
        spec = planes[:5].clone()
        spec[1 + min(bits, 4):] = 0                       # the library clears hop-bit planes 1..4 up front
        self._spec = spec

    def speculative_planes(self):
        return self._spec

    def finish(self):
        return self._hp


class FakePendingWithVerdict(FakePending):
    """engine.PendingBfs also offers verdict(): (deepest active level, CSR flags), exchanged with the planes."""

    def __init__(self, hops):
        super().__init__(hops)
        self._verdict = torch.tensor([int(hops.max()), 0], dtype=torch.int32)

    def verdict(self):
        return self._verdict

    def finish(self):
        raise AssertionError("finish() must not be needed when the verdicts travel with the planes")


def _worker(rank, world, port, k, result_dir, deep=True, with_verdict=False):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphpope_amd import distributed as pdist, synth
        from oracle import oracle
        ei, n = synth.rmat(9, edge_factor=4, seed=2)
        # rank-dependent depth: a long tail hangs off node 0 so that only some shards see large hop counts
        tail = np.arange(n, n + (40 if deep else 3))
        chain = np.stack([np.concatenate([[0], tail[:-1]]), tail])
        ei = np.concatenate([ei, chain, chain[::-1]], axis=1)
        n += len(tail)
        anchors = np.random.RandomState(7).choice(np.arange(n), k)
        anchors[0] = n - 1                                           # deep anchor lands in shard 0 only
        x = torch.from_numpy(np.random.RandomState(1).rand(n, 5).astype(np.float32))

        def bfs_fn(a):
            planes, bits = encode_planes(oracle.geodesic_hops(ei, n, a))
            return FakePlanes(planes, bits)

        calls = {"general": 0}

        def counted_bfs(a):
            calls["general"] += 1
            return bfs_fn(a)

        def copy_x(xx, ff, out):                                     # the feature copy that overlaps the exchange
            calls["copy_x"] = calls.get("copy_x", 0) + 1
            out[:, :ff] = xx

        def finalize_all(gathered, bits, nn, k_shard, xx, ff, out):
            assert xx is None                                        # the features were written by copy_x
            for g in range(gathered.shape[0]):
                numpy_finalize(gathered[g], bits, nn, k_shard, None, ff, out, g * k_shard)

        out = pdist.sharded_geodesic_features(x, n, anchors, None, counted_bfs, numpy_finalize, finalize_all_fn=finalize_all,
                                              begin_fn=lambda a: (FakePendingWithVerdict if with_verdict else FakePending)(
                                                  oracle.geodesic_hops(ei, n, a)), copy_x_fn=copy_x)
        assert calls["copy_x"] == (2 if deep else 1)
        # deep graph (> 15 hops somewhere): the speculative 4-bit exchange is rejected by ALL ranks and redone in general form
        assert calls["general"] == (1 if deep else 0), calls
        want = oracle.geodesic_features(x.numpy(), ei, n, anchors)
        ok = out.shape == (n, 5 + k) and out.is_contiguous() and np.array_equal(out.numpy().view(np.uint32), want.view(np.uint32))
        open(os.path.join(result_dir, f"rank{rank}"), "w").write("ok" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,k,deep,with_verdict", [(2, 128, True, False), (2, 7, True, True), (3, 10, True, True), (2, 1, True, False),
                                                       (2, 128, False, True), (3, 10, False, False), (2, 40, False, True),
                                                       # BASELINE.json configs[3] as written: 8 ranks x 128 anchors (two 64-anchor words
                                                       # per rank, an 8-shard expansion), and the ragged form (1 020 anchors: the last
                                                       # shard is 4 short and padded)
                                                       (8, 1024, False, True), (8, 1020, False, True), (8, 1020, True, False)])
def test_sharded_all_gather_reassembles_the_matrix(world, k, deep, with_verdict, tmp_path, oracle):
    mp.spawn(_worker, args=(world, _free_port(), k, str(tmp_path), deep, with_verdict), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}").read() == "ok"


def test_shard_bounds():
    from graphpope_amd import distributed as pdist
    a = np.arange(100, 110)
    assert pdist.shard_size(10, 3) == 4
    got = [pdist.shard_anchors(a, 3, r) for r in range(3)]
    assert [g[1] for g in got] == [4, 4, 2]
    assert got[0][0].tolist() == [100, 101, 102, 103] and got[2][0].tolist() == [108, 109, 109, 109]
    assert pdist.shard_anchors(np.array([5]), 2, 1)[0].tolist() == [5] and pdist.shard_anchors(np.array([5]), 2, 1)[1] == 0
    assert pdist.world_size() == 1 and pdist.rank() == 0


def test_plane_format_helper_matches_library_word_count():
    from graphpope_amd import _lib
    lib = _lib.load()
    for k in (1, 64, 65, 129, 256, 300, 1024):
        assert words_for(k) == lib.pope_words(k)
