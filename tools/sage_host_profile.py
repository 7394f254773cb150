#!/usr/bin/env python3
"""Host-side (enqueue) cost of each part of the SAGE training step (GPU box): the step is launch-bound, so this is where
the wall time goes.  cProfile over 200 steps, top entries by cumulative time."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, sage

dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
feats = torch.rand(n, 756, device=dev)
rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei_np[0], minlength=n))])
rng = np.random.default_rng(0)
batches = []
for b in range(8):
    seeds = rng.choice(n, 1550, replace=False)
    n_id, adjs = sage.sample_batch(rowptr, ei_np[1], seeds, sizes=(25, 10), rng=rng)
    batches.append((torch.as_tensor(n_id, device=dev), [a.to(dev) for a in adjs], torch.randint(0, 7, (1550,), device=dev)))
torch.manual_seed(0)
model = sage.SAGE(756, 7, 256, 3).to(dev)
from graphpope_amd.optim import Adam
opt = Adam(model.parameters(), lr=1e-3)
params = list(model.parameters())
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
def step(i):
    n_id, adjs, y = batches[i % 8]
    t = time.perf_counter(); x = feats.index_select(0, n_id); tick("index_select", t)
    t = time.perf_counter()
    for p_ in params: p_.grad = None
    tick("zero_grad", t)
    t = time.perf_counter(); out = model(x, adjs); tick("forward", t)
    t = time.perf_counter(); loss = F.cross_entropy(out, y); tick("loss", t)
    t = time.perf_counter(); loss.backward(); tick("backward", t)
    t = time.perf_counter(); opt.step(); tick("opt.step", t)
for i in range(20): step(i)
torch.cuda.synchronize(); T.clear()
N = 200
t0 = time.perf_counter()
for i in range(N): step(i)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print({k: round(v / N * 1e6, 1) for k, v in T.items()}, "host us/step", round(host / N * 1e6, 1))
pr = cProfile.Profile(); pr.enable()
for i in range(N): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
