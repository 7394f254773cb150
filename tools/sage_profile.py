#!/usr/bin/env python3
"""The SAGE training step under rocprofv3 (GPU box):
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -- python3 tools/sage_profile.py [eager|pool|sampler] [steps] [graph]
eager: bench.py's headline SAGE step (autograd on host-sized pre-sampled batches); pool / sampler: graphpope_amd.train.SageTrainStep.
One configuration per run, so that the per-kernel averages belong to it."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth  # noqa: E402
from graphpope_amd.optim import Adam  # noqa: E402
from graphpope_amd.sage import SAGE, sample_batch  # noqa: E402
from graphpope_amd.sampler import DeviceBatch, NeighborSampler  # noqa: E402
from graphpope_amd.train import SageTrainStep  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "pool"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
use_graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
dev = engine.require_gpu()
ei_np, n = synth.flickr_like(seed=1)
BATCH = 1550
feats = torch.rand((n, 756), device=dev)
torch.manual_seed(0)
model = SAGE(756, 7, 256, 3).to(dev)
opt = Adam(model.parameters(), lr=1e-3)
torch.autograd.set_multithreading_enabled(False)
if os.environ.get("GRAPHPOPE_STREAMK_XCD"):                    # A/B of the weight-gradient kernel's unit deal (POPE_KNOB_STREAMK_XCD)
    from graphpope_amd import _lib
    _lib.check(_lib.load().pope_debug_set(_lib.KNOB_STREAMK_XCD, int(os.environ["GRAPHPOPE_STREAMK_XCD"])))
if mode == "eager":
    from graphpope_amd.sage import IndexedFeatures, cross_entropy
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei_np[0], minlength=n))])
    rng = np.random.default_rng(0)
    batches = []
    for b in range(8):
        seeds = rng.choice(n, BATCH, replace=False)
        n_id, adjs = sample_batch(rowptr, ei_np[1], seeds, sizes=(25, 10), rng=rng)
        batches.append((torch.as_tensor(n_id, device=dev), [a.to(dev) for a in adjs], torch.randint(0, 7, (BATCH,), device=dev)))
    params = list(model.parameters())
    one = torch.ones((), device=dev)

    def step(i):
        n_id, adjs, y = batches[i % 8]
        for p in params:
            p.grad = None
        loss = cross_entropy(model(IndexedFeatures(feats, n_id), adjs), y, unit_upstream=True, loss_in=opt)
        loss.backward(gradient=one)
        opt.step()
elif mode == "pool":
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei_np[0], minlength=n))])
    rng = np.random.default_rng(0)
    pool = []
    for b in range(4):
        seeds = rng.choice(n, BATCH, replace=False)
        n_id, adjs = sample_batch(rowptr, ei_np[1], seeds, sizes=(25, 10), rng=rng)
        db = DeviceBatch(BATCH, (25, 10), dev)
        db.load(torch.as_tensor(n_id, device=dev), [a.to(dev) for a in adjs])
        pool.append((db, torch.randint(0, 7, (BATCH,), device=dev)))
    st = SageTrainStep(model, opt, feats, BATCH, (25, 10), sampler=None, graph=use_graph)

    def step(i):
        db, y = pool[i % 4]
        st.load_batch(db, y)
        st.run()
else:
    csr = engine.build_csr(torch.as_tensor(ei_np, device=dev), n)
    sampler = NeighborSampler(csr.rowptr, csr.col, n, (25, 10))
    perm = torch.randperm(n, device=dev)
    labels = torch.randint(0, 7, (n,), device=dev)
    st = SageTrainStep(model, opt, feats, BATCH, sampler=sampler, graph=use_graph)

    def step(i):
        lo = (i * BATCH) % (n - BATCH)
        sd = perm[lo:lo + BATCH]
        st.step(sd, labels[sd])
for i in range(steps):
    step(i)
torch.cuda.synchronize()
print("done", mode, steps, "graph" if use_graph else "eager")
