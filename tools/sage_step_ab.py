#!/usr/bin/env python3
"""A/B of the SAGE training step on the bench's Flickr-shaped batches (GPU box): python tools/sage_step_ab.py [steps]
eager host-sized / eager device-extent / replayed graph, side lanes on / off, pre-sampled pool / sampler in the step."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth  # noqa: E402
from graphpope_amd.optim import Adam  # noqa: E402
from graphpope_amd.sage import SAGE, IndexedFeatures, cross_entropy, sample_batch  # noqa: E402
from graphpope_amd.sampler import DeviceBatch, NeighborSampler  # noqa: E402
from graphpope_amd.train import SageTrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
only = sys.argv[2] if len(sys.argv) > 2 else ""
lib = _lib.load()
dev = engine.require_gpu()
ei_np, n = synth.flickr_like(seed=1)
BATCH, HIDDEN = 1550, 256
feats = torch.rand((n, 756), device=dev)
rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei_np[0], minlength=n))])
rng = np.random.default_rng(0)
batches = []
for b in range(8):
    seeds = rng.choice(n, BATCH, replace=False)
    n_id, adjs = sample_batch(rowptr, ei_np[1], seeds, sizes=(25, 10), rng=rng)
    batches.append((torch.as_tensor(n_id, device=dev), [a.to(dev) for a in adjs], torch.randint(0, 7, (BATCH,), device=dev)))
pool = []
for n_id, adjs, y in batches:
    db = DeviceBatch(BATCH, (25, 10), dev)
    db.load(n_id, adjs)
    pool.append((db, y))
csr = engine.build_csr(torch.as_tensor(ei_np, device=dev), n)
sampler = NeighborSampler(csr.rowptr, csr.col, n, (25, 10))
perm = torch.randperm(n, device=dev)
labels = torch.randint(0, 7, (n,), device=dev)
one = torch.ones((), device=dev)


def fresh():
    torch.manual_seed(0)
    m = SAGE(756, 7, HIDDEN, 3).to(dev)
    return m, Adam(m.parameters(), lr=1e-3)


def timeit(fn, label):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    print(f"{label:60s} {(time.perf_counter() - t0) / steps * 1e3:7.3f} ms/step", flush=True)


torch.autograd.set_multithreading_enabled(False)
for lanes in (0,):
    lib.pope_debug_set(_lib.KNOB_SAGE_LANES, lanes)
    tag = f"lanes={lanes} "
    if not only or only == "eager":
        m, opt = fresh()
        params = list(m.parameters())

        def eager(i):
            n_id, adjs, y = batches[i % 8]
            for p in params:
                p.grad = None
            loss = cross_entropy(m(IndexedFeatures(feats, n_id), adjs), y)
            loss.backward(gradient=one)
            opt.step()
        timeit(eager, tag + "eager, host-sized pre-sampled batches")
    for use_graph in (False, True):
        if only and only != "train":
            continue
        m, opt = fresh()
        st = SageTrainStep(m, opt, feats, BATCH, (25, 10), sampler=None, graph=use_graph)

        def pooled(i):
            db, y = pool[i % 8]
            st.load_batch(db, y)
            st.run()
        timeit(pooled, tag + f"SageTrainStep graph={use_graph}, pre-sampled pool (device extents)")
        m, opt = fresh()
        st2 = SageTrainStep(m, opt, feats, BATCH, sampler=sampler, graph=use_graph)

        def sampled(i):
            lo = (i * BATCH) % (n - BATCH)
            sd = perm[lo:lo + BATCH]
            st2.step(sd, labels[sd])
        timeit(sampled, tag + f"SageTrainStep graph={use_graph}, sampler inside the step")
        if not use_graph:
            m, opt = fresh()
            st3 = SageTrainStep(m, opt, feats, BATCH, sampler=sampler, graph=False, prefetch=True)

            def ahead(i):
                lo, lo2 = (i * BATCH) % (n - BATCH), ((i + 1) * BATCH) % (n - BATCH)
                sd, sd2 = perm[lo:lo + BATCH], perm[lo2:lo2 + BATCH]
                st3.step(sd, labels[sd], sd2, labels[sd2])
            timeit(ahead, tag + "SageTrainStep eager, next batch sampled on a side stream")
