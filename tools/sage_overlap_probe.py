#!/usr/bin/env python3
"""Probe (GPU box): does the fan-out sampler hide beside the training step's compute inside ONE replayed HIP graph?
Captures  [sample batch B on a second stream]  ||  [forward + loss + backward + Adam on batch A]  and, for comparison, the two in
sequence and the compute alone.  Batch A is sampled once before the capture, so the arithmetic of every replay is the same."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth  # noqa: E402
from graphpope_amd.optim import Adam  # noqa: E402
from graphpope_amd.sage import SAGE  # noqa: E402
from graphpope_amd.sampler import DeviceBatch, NeighborSampler  # noqa: E402
from graphpope_amd.train import SageTrainStep, copy_segments  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = engine.require_gpu()
ei_np, n = synth.flickr_like(seed=1)
BATCH = 1550
feats = torch.rand((n, 756), device=dev)
csr = engine.build_csr(torch.as_tensor(ei_np, device=dev), n)
perm = torch.randperm(n, device=dev)
labels = torch.randint(0, 7, (n,), device=dev)
torch.autograd.set_multithreading_enabled(False)
for mode in ("compute", "serial", "parallel", "parallel", "serial", "compute"):
    torch.manual_seed(0)
    model = SAGE(756, 7, 256, 3).to(dev)
    opt = Adam(model.parameters(), lr=1e-3)
    sampler = NeighborSampler(csr.rowptr, csr.col, n, (25, 10))
    st = SageTrainStep(model, opt, feats, BATCH, sampler=sampler, graph=False)
    other = DeviceBatch(BATCH, sampler.sizes, dev)
    seeds_a, seeds_b = perm[:BATCH].contiguous(), perm[BATCH:2 * BATCH].contiguous()
    side, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def body():
        cur = torch.cuda.current_stream()
        if mode == "parallel":
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                sampler.sample_device(seeds_b, seed=0, out=other, seed_dev=st.state.sample_seed)
        elif mode == "serial":
            sampler.sample_device(seeds_b, seed=0, out=other, seed_dev=st.state.sample_seed)
        st._body(sample=False)
        if mode == "parallel":
            cur.wait_stream(s2)

    copy_segments([st.seeds, st.y], [seeds_a, labels[seeds_a]])
    sampler.sample_device(st.seeds, seed=0, out=st.batch, seed_dev=st.state.sample_seed)
    st.model.train(True)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        body()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    print(f"{mode:9s} {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per replay")
