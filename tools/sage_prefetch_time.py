#!/usr/bin/env python3
"""The SAGE training step with the fan-out sampled on the GPU (GPU box): three ways of getting the batch, same model and sampler.
    graph     the sampler inside the replayed step (bench.py's sage_sampled_ms_per_step)
    eager     the sampler inside the step, eager launches, device extents
    prefetch  SageTrainStep(prefetch=True): the NEXT batch sampled on a side stream while this step computes (eager launches)
python3 tools/sage_prefetch_time.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth  # noqa: E402
from graphpope_amd.optim import Adam  # noqa: E402
from graphpope_amd.sage import SAGE  # noqa: E402
from graphpope_amd.sampler import NeighborSampler  # noqa: E402
from graphpope_amd.train import SageTrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = engine.require_gpu()
ei_np, n = synth.flickr_like(seed=1)
BATCH = 1550
feats = torch.rand((n, 756), device=dev)
csr = engine.build_csr(torch.as_tensor(ei_np, device=dev), n)
perm = torch.randperm(n, device=dev)
labels = torch.randint(0, 7, (n,), device=dev)
torch.autograd.set_multithreading_enabled(False)
for mode in ("graph", "eager", "prefetch", "prefetch", "graph"):
    torch.manual_seed(0)
    model = SAGE(756, 7, 256, 3).to(dev)
    opt = Adam(model.parameters(), lr=1e-3)
    sampler = NeighborSampler(csr.rowptr, csr.col, n, (25, 10))
    st = SageTrainStep(model, opt, feats, BATCH, sampler=sampler, graph=mode == "graph", prefetch=mode == "prefetch")
    batches = [(perm[(i * BATCH) % (n - BATCH):(i * BATCH) % (n - BATCH) + BATCH].contiguous(),) for i in range(steps + 12)]
    batches = [(s[0], labels[s[0]]) for s in batches]

    def run(i):
        sd, y = batches[i]
        if mode == "prefetch":
            nsd, ny = batches[i + 1]
            st.step(sd, y, nsd, ny)
        else:
            st.step(sd, y)
    for i in range(8):
        run(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(8, 8 + steps):
        run(i)
    torch.cuda.synchronize()
    print(f"{mode:9s} {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per step   loss {float(st.loss):.4f}")
