"""BASELINE.json configs[3] and configs[4] at full size on one MI355X, through the C ABI, against the C oracle.

configs[3]: Flickr-shaped graph, 1 024 anchors (np.random seed 42: 1 020 distinct) -- unsharded, and sharded over a
            world-2 process group (512 anchors per rank, hop planes all-gathered) with the real kernels.
configs[4]: R-MAT scale 22 (4.2 M nodes, ~65 M CSR slots), 512 anchors -- eight seeded columns bit-exact against the
            oracle, plus the size-independent properties of a hop matrix on every column.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu

F = 500                    # /root/reference/main.py:77-79


@pytest.fixture(scope="module")
def dev():
    from graphpope_amd import engine
    return engine.require_gpu()


def test_config3_flickr_1024_anchors_unsharded(dev, oracle):
    from graphpope_amd import engine, synth
    ei, n = synth.flickr_like()
    anchors = synth.seeded_anchors(n, 1024, 42)
    assert len(np.unique(anchors)) == 1020                                # SURVEY.md §7 trap 3: duplicates stay, in draw order
    x = torch.rand((n, F), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    out, hp = engine.geodesic_run(x, torch.as_tensor(ei, device=dev), n, anchors)
    torch.cuda.synchronize()
    want = oracle.geodesic_hops(ei, n, anchors)
    assert np.array_equal(engine.hop_matrix(hp).cpu().numpy(), want)
    assert torch.equal(out[:, :F], x)
    assert np.array_equal(out[:, F:].cpu().numpy().view(np.uint32), oracle.hops_to_embedding(want).view(np.uint32))
    # duplicate anchors give identical columns
    first = {}
    for j, a in enumerate(anchors.tolist()):
        if a in first:
            assert torch.equal(out[:, F + j], out[:, F + first[a]])
        first.setdefault(a, j)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _config3_worker(rank, world, port, result_dir, k=1024):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from graphpope_amd import engine, synth
        from oracle import oracle
        ei, n = synth.flickr_like()
        anchors = synth.seeded_anchors(n, 1024, 42)[:k]
        x = torch.rand((n, 16), generator=torch.Generator().manual_seed(1))
        out = engine.geodesic_features(x.cuda(), torch.as_tensor(ei).cuda(), n, anchors)      # ceil(k / world) anchors on this rank
        want = oracle.geodesic_features(x.numpy(), ei, n, anchors)
        ok = tuple(out.shape) == (n, 16 + k) and np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        open(os.path.join(result_dir, f"rank{rank}"), "w").write("ok" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


def test_config3_flickr_1024_anchors_sharded_world2(tmp_path, oracle):
    """engine.geodesic_features inside a world-2 group: anchor shards, plane all-gather, one-pass expansion of both shards."""
    mp.spawn(_config3_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(tmp_path / f"rank{r}").read() == "ok"


def test_config3_ragged_1020_anchors_sharded_world4(tmp_path, oracle):
    """Four ranks on the one card (gloo carries the planes, the real kernels do the rest): 1 020 anchors -> 255 per rank,
    W = 4 words per shard with a ragged last word, 4-shard expansion with the padded tail dropped."""
    mp.spawn(_config3_worker, args=(4, _free_port(), str(tmp_path), 1020), nprocs=4, join=True)
    for r in range(4):
        assert open(tmp_path / f"rank{r}").read() == "ok"


def test_config4_rmat22_512_anchors(dev, oracle):
    from graphpope_amd import engine, synth
    ei, n = synth.rmat(22, edge_factor=8, seed=1)
    assert n == 4_194_304 and ei.shape[1] > 60_000_000
    anchors = synth.seeded_anchors(n, 512, 42)
    eid = torch.as_tensor(ei, device=dev)
    out, hp = engine.geodesic_run(None, eid, n, anchors)                   # F = 0 in this config: the [N, 512] embedding
    hops = engine.hop_matrix(hp)                                           # int32 [N, 512] on the device (8.6 GB)
    torch.cuda.synchronize()
    # eight seeded columns against the oracle, integers and float32 bit patterns
    cols = np.random.RandomState(0).choice(512, 8, replace=False)
    cd = torch.as_tensor(cols, device=dev)
    want = oracle.geodesic_hops(ei, n, anchors[cols])
    assert np.array_equal(hops[:, cd].cpu().numpy(), want)
    assert np.array_equal(out[:, cd].cpu().numpy().view(np.uint32), oracle.hops_to_embedding(want).view(np.uint32))
    # every anchor is at distance 0 from itself
    ad = torch.as_tensor(anchors, device=dev)
    assert bool((hops[ad, torch.arange(512, device=dev)] == 0).all())
    # the embedding is the hop matrix: 1 / (h + 1), 0 where unreachable (exact in float32 for these small integers)
    for c0 in range(0, 512, 64):
        h = hops[:, c0:c0 + 64]
        e = torch.where(h >= 0, 1.0 / (h.to(torch.float32) + 1.0), torch.zeros((), device=dev))
        assert torch.equal(out[:, c0:c0 + 64], e)
    # along every edge of this symmetric graph the two ends are both reachable or both not, and their hops differ by <= 1
    src, dst = eid[0], eid[1]
    for c0 in range(0, 512, 32):
        hu, hv = hops[:, c0:c0 + 32][src], hops[:, c0:c0 + 32][dst]
        assert bool(((hu >= 0) == (hv >= 0)).all())
        assert bool(((hu - hv).abs() <= 1).all())
        del hu, hv
    assert 0 < hp.max_hop < 16 and int(hops.max()) == hp.max_hop
