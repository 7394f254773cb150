#!/usr/bin/env python3
"""Pure host-side cost of each op of the SAGE training step (GPU box): the same Python / ctypes / autograd path on tiny
tensors, so the GPU never back-pressures the launch queue.  Microseconds per call."""
import os, sys, time
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, sage
from graphpope_amd.optim import Adam
dev = engine.require_gpu()
n_dst, n_src, c = 64, 128, 32
rowptr = torch.arange(0, n_dst + 1, dtype=torch.int32) * 2
adj0 = sage.SampledAdj(rowptr, torch.randint(0, n_src, (2 * n_dst,), dtype=torch.int32), n_src).to(dev)
adj1 = sage.SampledAdj(rowptr[:17], torch.randint(0, n_dst, (32,), dtype=torch.int32), n_dst).to(dev)
model = sage.SAGE(c, 3, c, 3).to(dev)
opt = Adam(model.parameters(), lr=1e-3)
topt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)
params = list(model.parameters())
x = torch.randn(n_src, c, device=dev); y = torch.randint(0, 3, (16,), device=dev)
feats = torch.randn(1000, c, device=dev); idx = torch.randint(0, 1000, (n_src,), device=dev)
def bench(name, fn, reps=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    print(f"{name:34s} {dt:7.1f} us")
    return dt
def step_parts():
    out = model(x, [adj0, adj1]); loss = F.cross_entropy(out, y); loss.backward()
bench("index_select", lambda: feats.index_select(0, idx))
bench("forward (2 conv + bn)", lambda: model(x, [adj0, adj1]))
def fwd_loss():
    return F.cross_entropy(model(x, [adj0, adj1]), y)
bench("forward + loss", fwd_loss)
def fb():
    for p in params: p.grad = None
    fwd_loss().backward()
t_fb = bench("forward + loss + backward", fb)
def full():
    fb(); opt.step()
t_full = bench("... + graphpope_amd.optim.Adam", full)
def full_t():
    fb(); topt.step()
bench("... + torch Adam(fused=True)", full_t)
conv = model.convs[0]
bench("SAGEConv forward only (no grad)", lambda: conv((x, x[:n_dst]), adj0))
with torch.no_grad():
    bench("SAGEConv forward under no_grad", lambda: conv((x, x[:n_dst]), adj0))
bench("bn_relu_dropout forward", lambda: sage.bn_relu_dropout(x, model.bns[0], 0.5, True))
bench("torch.empty", lambda: torch.empty(10, device=dev))
with torch.autograd.set_multithreading_enabled(False):
    bench("fwd + loss + bwd, single-threaded autograd", fb)
    bench("... + graphpope_amd Adam, single-threaded", full)
