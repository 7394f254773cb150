#!/usr/bin/env python3
"""Cost of one device-side mini-batch draw (GPU box): wall per sample() call; run under rocprofv3 for the kernel list."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth
from graphpope_amd.sampler import NeighborSampler
dev = engine.require_gpu()
ei, n = synth.flickr_like()
csr = engine.build_csr(torch.as_tensor(ei, device=dev), n)
s = NeighborSampler(csr.rowptr, csr.col, n, (25, 10))
perm = torch.randperm(n, device=dev)
for i in range(10): s.sample(perm[i * 1550:(i + 1) * 1550], seed=i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50): n_id, adjs = s.sample(perm[i * 1550:(i + 1) * 1550], seed=i)
torch.cuda.synchronize()
print(json.dumps({"us_per_batch": (time.perf_counter() - t0) / 50 * 1e6, "n_src": int(n_id.numel()), "nnz": [int(a.col.numel()) for a in adjs]}))
