"""The sharded path with the REAL HIP kernels: two ranks (gloo, both on cuda:0) and one NCCL rank (RCCL all-gather call)."""
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
