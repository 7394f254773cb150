"""The sharded path with the REAL HIP kernels: two ranks (gloo, both on cuda:0), and the RCCL collectives on one rank.

A one-GPU box cannot hold two RCCL ranks, so what runs here on the "nccl" backend is world 1: the all_gather_into_tensor
of the hop planes, the 8-byte verdict all-gather and the speculative finalize_shards on NCCL-gathered planes all execute
once on the GPU (test_nccl_collectives_run_on_one_rank).  RCCL with world > 1 is covered by the driver's multi-GPU
bench only (SCALE_rNN.json); DESIGN.md says so."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, backend, k, result_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from graphpope_amd import engine, synth
        from oracle import oracle
        ei, n = synth.rmat(12, edge_factor=8, seed=4)
        anchors = np.random.RandomState(3).choice(np.arange(n), k)
        x = torch.rand(n, 8, generator=torch.Generator().manual_seed(1))
        out = engine.geodesic_features(x.cuda(), torch.as_tensor(ei).cuda(), n, anchors)     # shards: group is initialised
        want = oracle.geodesic_features(x.numpy(), ei, n, anchors)
        ok = tuple(out.shape) == (n, 8 + k) and np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        open(os.path.join(result_dir, f"rank{rank}"), "w").write("ok" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,backend,k", [(2, "gloo", 256), (2, "gloo", 37), (1, "nccl", 96)])
def test_sharded_path_on_the_gpu(world, backend, k, tmp_path, oracle):
    mp.spawn(_worker, args=(world, _free_port(), backend, k, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"rank{r}").read() == "ok"


def _nccl_worker(rank, world, port, k, general, result_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from graphpope_amd import engine, synth
        from graphpope_amd import distributed as pdist
        from oracle import oracle
        ei, n = synth.rmat(12, edge_factor=8, seed=4)
        anchors = np.random.RandomState(3).choice(np.arange(n), k)
        x = torch.rand(n, 8, generator=torch.Generator().manual_seed(1)).cuda()
        csr = engine.build_csr(torch.as_tensor(ei).cuda(), n, defer_check=True)
        calls = {"begin": 0, "all": 0}

        def begin(a):
            calls["begin"] += 1
            return engine.PendingBfs(csr, a)

        def finalize_all(*a):
            calls["all"] += 1
            return engine.finalize_shards(*a)
        # engine.geodesic_features short-cuts world == 1; call the sharded path itself so that the collectives run
        out = pdist.sharded_geodesic_features(
            x, n, anchors, None, bfs_fn=lambda a: engine.bfs(csr, a), finalize_fn=engine.finalize,
            finalize_all_fn=finalize_all, begin_fn=None if general else begin, copy_x_fn=engine.copy_features)
        want = oracle.geodesic_features(x.cpu().numpy(), ei, n, anchors)
        ok = tuple(out.shape) == (n, 8 + k) and np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        ok = ok and calls["all"] == 1 and calls["begin"] == (0 if general else 1)
        open(os.path.join(result_dir, f"rank{rank}"), "w").write("ok" if ok else "MISMATCH")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("k,general", [(96, False), (130, False), (96, True)])
def test_nccl_collectives_run_on_one_rank(k, general, tmp_path, oracle):
    """distributed.sharded_geodesic_features on the RCCL backend (world 1): all_gather_into_tensor of the planes, the verdict
    all-gather, the feature copy underneath, finalize_shards on the gathered buffer (speculative path), and the
    all_reduce(MAX) + all-gather of the general path -- bit-exact against the oracle."""
    mp.spawn(_nccl_worker, args=(1, _free_port(), k, general, str(tmp_path)), nprocs=1, join=True)
    assert open(tmp_path / "rank0").read() == "ok"
