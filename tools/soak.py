#!/usr/bin/env python3
"""3 000 back-to-back geodesic and node2vec calls with fresh anchors (GPU box): host memory, device memory and file descriptors
must stay flat (per-call slots, the side stream and its events are created once)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphpope_amd import engine, synth
import psutil
dev = engine.require_gpu()
ei_np, n = synth.rmat(14, edge_factor=8, seed=3)
ei = torch.as_tensor(ei_np, device=dev)
x = torch.rand(n, 64, device=dev)
emb = torch.randn(n, 128, device=dev)
proc = psutil.Process()
rs = np.random.RandomState(0)
t0 = time.time()
for it in range(3000):
    anchors = rs.choice(n, 128)
    out, hp = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)
    o2 = engine.pairwise_features(x, emb, anchors, "euclidean")
    if it % 500 == 0:
        torch.cuda.synchronize()
        print(it, "rss MB %.0f" % (proc.memory_info().rss / 1e6), "cuda MB %.0f" % (torch.cuda.memory_allocated() / 1e6), "fds", proc.num_fds(), flush=True)
torch.cuda.synchronize()
print("done in %.1f s" % (time.time() - t0), "rss MB %.0f" % (proc.memory_info().rss / 1e6), "fds", proc.num_fds())
