#!/usr/bin/env python3
"""Would the geodesic step be faster as a replayed HIP graph?  (GPU box): python tools/graph_gap_experiment.py
The same launches -- CSR build, 12 enqueued BFS levels (pope_geodesic_bfs_begin), finalise with 4 hop bits -- enqueued on a
stream every time against captured once and replayed.  (pope_geodesic_run itself cannot be captured: it waits for the
verdict on the host.)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth
dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
F, K = 500, 256
anchors = synth.seeded_anchors(n, K, 42)
ei = torch.as_tensor(ei_np, device=dev)
x = torch.rand(n, F, device=dev)
out = torch.empty(n, F + K, device=dev)
csr_hold = {}


def body():
    csr = engine.build_csr(ei, n, defer_check=True)
    pb = engine.PendingBfs(csr, anchors)
    engine.finalize(pb.speculative_planes(), 4, n, K, x, F, out, 0)
    csr_hold["c"], csr_hold["p"] = csr, pb             # keep the buffers alive (the graph refers to them)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        body()
side.synchronize()
print(f"stream launches (12 levels enqueued, no verdict wait): {timeit(body):.4f} ms per step")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    body()
print(f"captured once, replayed:                               {timeit(g.replay):.4f} ms per step")
ref = out.clone()
g.replay()
torch.cuda.synchronize()
print("replay reproduces the result:", bool(torch.equal(ref, out)))
step = lambda: engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)
print(f"pope_geodesic_run (10 levels, one verdict wait):       {timeit(step):.4f} ms per step")
