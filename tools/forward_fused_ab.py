#!/usr/bin/env python3
"""Layer-0 forward orders on the bench's Flickr-shaped batches (GPU box): python tools/forward_fused_ab.py [steps]
POPE_KNOB_SAGE_FORWARD_OVERLAP 1 (gather beside half of the projection + a second launch), 0 (one after the other),
2 (one fused launch, tiles wait for their rows): the layer alone between HIP events, then the whole training step."""
import ctypes
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
orders = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 0, 2, 1]
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
torch.autograd.set_multithreading_enabled(False)


def layer_alone(order):
    n_id, adjs, _ = batches[0]
    a0 = adjs[0]
    c_in, c_out = 756, 256
    g = torch.Generator().manual_seed(5)
    w_l, w_r = (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev), (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev)
    b = torch.randn(c_out, generator=g).to(dev)
    agg = torch.empty((a0.n_dst, c_in), device=dev)
    x_dst = torch.empty((a0.n_dst, c_in), device=dev)
    out = torch.empty((a0.n_dst, c_out), device=dev)
    scratch = torch.empty(max(lib.sage_conv_forward_scratch_bytes(a0.n_dst, c_in, c_out), 16), dtype=torch.uint8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, order)

    def call():
        _lib.check(lib.sage_conv_forward_indexed(_lib.ptr(a0.rowptr), _lib.ptr(a0.col), _lib.ptr(n_id), a0.n_src, a0.n_dst, a0.col.numel(),
                                                 _lib.ptr(feats), n, c_in, _lib.ptr(w_l), _lib.ptr(b), _lib.ptr(w_r), c_out, _lib.ptr(agg),
                                                 _lib.ptr(x_dst), _lib.ptr(out), _lib.ptr(scratch), scratch.numel(), None, stream))
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        call()
    e1.record()
    torch.cuda.synchronize()
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
    return e0.elapsed_time(e1) / steps * 1e3, out


def timeit(fn):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def step_times(order):
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, order)
    torch.manual_seed(0)
    m = SAGE(756, 7, HIDDEN, 3).to(dev)
    opt = Adam(m.parameters(), lr=1e-3)
    params = list(m.parameters())

    def eager(i):
        n_id, adjs, y = batches[i % 8]
        for p in params:
            p.grad = None
        loss = cross_entropy(m(IndexedFeatures(feats, n_id), adjs), y)
        loss.backward(gradient=one)
        opt.step()
    t_eager = timeit(eager)
    torch.manual_seed(0)
    m = SAGE(756, 7, HIDDEN, 3).to(dev)
    opt = Adam(m.parameters(), lr=1e-3)
    st = SageTrainStep(m, opt, feats, BATCH, sampler=sampler, graph=True)

    def sampled(i):
        lo = (i * BATCH) % (n - BATCH)
        sd = perm[lo:lo + BATCH]
        st.step(sd, labels[sd])
    t_graph = timeit(sampled)
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
    return t_eager, t_graph


ref = None
for order in orders:
    us, out = layer_alone(order)
    if ref is None:
        ref = out.clone()
    err = float((out - ref).abs().max()) / float(ref.abs().max())
    print(f"order {order}: layer-0 forward alone {us:7.1f} us   (max |diff| / max |out| vs the first order: {err:.2e})", flush=True)
for order in orders:
    te, tg = step_times(order)
    print(f"order {order}: eager pre-sampled step {te:7.4f} ms   replayed graph with the sampler inside {tg:7.4f} ms", flush=True)
