#!/usr/bin/env python3
"""A/B of two builds of the library on the BFS (GPU box): python tools/ab_lib.py tools/_diag/libgraphpope_hip_nt.so
Each build runs in its own subprocess (one library per process), interleaved, same device."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, time, json
sys.path.insert(0, %r)
from graphpope_amd import _lib
if sys.argv[1] != "default": _lib.LIB_PATH = sys.argv[1]
import torch, numpy as np
from graphpope_amd import engine, synth
dev = engine.require_gpu()
ei, n = synth.flickr_like(); anchors = synth.seeded_anchors(n, 256, 42)
eid = torch.as_tensor(ei, device=dev)
csr = engine.build_csr(eid, n)
for _ in range(5): engine.bfs(csr, anchors)
torch.cuda.synchronize(); ts = []
for _ in range(30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); engine.bfs(csr, anchors); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
print(json.dumps({"median_us": float(np.median(ts)), "min_us": float(min(ts))}))
''' % ROOT
for rnd in range(4):
    for lib in ["default"] + sys.argv[1:]:
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        print(rnd, os.path.basename(lib), out.stdout.strip().split("\n")[-1] if out.returncode == 0 else out.stderr[-300:])
