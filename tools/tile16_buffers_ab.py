#!/usr/bin/env python3
"""Whole-tile forward GEMM with 3 or 4 LDS stage buffers (POPE_KNOB_GEMM_TILE16_BUFFERS) on the bench's layer-0 block (GPU box):
the layer alone in both launch orders, outputs compared bit for bit (the arithmetic is the same), then the training step."""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
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
one = torch.ones((), device=dev)
torch.autograd.set_multithreading_enabled(False)


def layer_alone(order, bufs):
    n_id, adjs, _ = batches[0]
    a0 = adjs[0]
    c_in, c_out = 756, 256
    g = torch.Generator().manual_seed(5)
    w_l, w_r = (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev), (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev)
    b = torch.randn(c_out, generator=g).to(dev)
    agg = torch.empty((a0.n_dst, c_in), device=dev)
    x_dst = torch.empty((a0.n_dst, c_in), device=dev)
    out = torch.full((a0.n_dst, c_out), -7.0, device=dev)
    scratch = torch.empty(max(lib.sage_conv_forward_scratch_bytes(a0.n_dst, c_in, c_out), 16), dtype=torch.uint8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, order)
    lib.pope_debug_set(_lib.KNOB_GEMM_TILE16_BUFFERS, bufs)

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
    lib.pope_debug_set(_lib.KNOB_GEMM_TILE16_BUFFERS, 4)
    return e0.elapsed_time(e1) / steps * 1e3, out


def step_time(bufs):
    lib.pope_debug_set(_lib.KNOB_GEMM_TILE16_BUFFERS, bufs)
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
    for i in range(6):
        eager(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        eager(i)
    torch.cuda.synchronize()
    lib.pope_debug_set(_lib.KNOB_GEMM_TILE16_BUFFERS, 4)
    return (time.perf_counter() - t0) / steps * 1e3


for order in (0, 1):
    outs = {}
    for bufs in (3, 4, 3, 4):
        us, out = layer_alone(order, bufs)
        outs[bufs] = out
        print(f"order {order} buffers {bufs}: layer-0 forward alone {us:7.1f} us", flush=True)
    print(f"order {order}: outputs of 3 and 4 buffers bit-identical: {torch.equal(outs[3], outs[4])}", flush=True)
for bufs in (3, 4, 3, 4):
    print(f"buffers {bufs}: eager pre-sampled step {step_time(bufs):7.4f} ms", flush=True)
