#!/usr/bin/env python3
"""Where the host -> host Graphpope call spends its time (GPU box): python tools/boundary_breakdown.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth
dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
F, K = 500, 256
x = torch.rand(n, F)
ei_cpu = torch.as_tensor(ei_np)
anchors = synth.seeded_anchors(n, K, 42)
def sync(): torch.cuda.current_stream().synchronize()
def t(fn, reps=7):
    ts = []
    for _ in range(reps):
        sync(); t0 = time.perf_counter(); r = fn(); sync(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[2:])), r
ms, ei = t(lambda: engine.stage_to_device(ei_cpu, dev)); print(f"edge_index host -> pinned -> device (14.4 MB): {ms:.3f} ms")
ms, emb = t(lambda: engine.geodesic_features(None, ei, n, anchors, shard=False)); print(f"CSR + BFS + [N, K] expansion on the GPU:     {ms:.3f} ms")
out = torch.empty((n, F + K), dtype=torch.float32, pin_memory=True)
ms, _ = t(lambda: engine.copy_columns_to_host(emb, out[:, F:])); print(f"pitched D2H of [N, K] (91.4 MB):            {ms:.3f} ms  -> {91.392e6 / ms / 1e6:.1f} GB/s")
lin = torch.empty((n, K), dtype=torch.float32, pin_memory=True)
ms, _ = t(lambda: lin.copy_(emb, non_blocking=True)); print(f"contiguous D2H of the same bytes:            {ms:.3f} ms  -> {91.392e6 / ms / 1e6:.1f} GB/s")
for th in (4, 8, 16, 32):
    ms, _ = t(lambda: engine.host_copy_2d(x, out[:, :F], threads=th)); print(f"host copy of data.x (178.5 MB), {th:2d} threads:  {ms:.3f} ms  -> {2 * 178.5e6 / ms / 1e6:.1f} GB/s (read + write)")
ms, _ = t(lambda: torch.empty((n, F + K), dtype=torch.float32, pin_memory=True)); print(f"pinned allocation of the result (cached):    {ms:.3f} ms")
def whole():
    o = torch.empty((n, F + K), dtype=torch.float32, pin_memory=True)
    e = engine.geodesic_features(None, engine.stage_to_device(ei_cpu, dev), n, anchors, shard=False)
    engine.copy_columns_to_host(e, o[:, F:]); engine.host_copy_2d(x, o[:, :F]); return o
ms, _ = t(whole); print(f"the whole call body:                         {ms:.3f} ms")
