#!/usr/bin/env python3
"""What plain copies reach on this box (GPU box): contiguous 224 MB copy, and torch's strided out[:, :F] = x, HBM-cold."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine
dev = engine.require_gpu()
n, f, k = 89250, 500, 256
x = torch.rand((n, f), device=dev)
out = torch.empty((n, f + k), device=dev)
src = torch.rand(n * (f + k + f) // 2, device=dev)      # same total traffic as finalise: (2F + K) * 4 * N bytes moved
dst = torch.empty_like(src)
evict = torch.empty(512 * 1024 * 1024 // 4, device=dev)
def timed(fn, reps=10):
    ts = []
    for _ in range(reps):
        evict.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts[2:]))
t = timed(lambda: dst.copy_(src))
print(f"contiguous copy of {src.numel() * 4 / 1e6:.0f} MB: {t:.1f} us -> {2 * src.numel() * 4 / t / 1e3:.0f} GB/s")
t = timed(lambda: out[:, :f].copy_(x))
print(f"strided out[:, :F] = x ({x.numel() * 4 / 1e6:.0f} MB): {t:.1f} us -> {2 * x.numel() * 4 / t / 1e3:.0f} GB/s")
t = timed(lambda: out.fill_(0.5))
print(f"fill of out ({out.numel() * 4 / 1e6:.0f} MB): {t:.1f} us -> {out.numel() * 4 / t / 1e3:.0f} GB/s")
t = timed(lambda: x.sum())
print(f"read of x ({x.numel() * 4 / 1e6:.0f} MB): {t:.1f} us -> {x.numel() * 4 / t / 1e3:.0f} GB/s")
