#!/usr/bin/env python3
"""configs[1] BFS alone, five times, through pope_geodesic_bfs_begin / _finish (always 12 level launches per BFS; the
one-call forms size their run by the previous depth): the workload of tools/pmc_expand.sh, whose per-level counter
averages go by launch position modulo 12."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth
dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
anchors = synth.seeded_anchors(n, 256, 42)
csr = engine.build_csr(torch.as_tensor(ei_np, device=dev), n)
for _ in range(5):
    hp = engine.PendingBfs(csr, anchors).finish()
torch.cuda.synchronize()
print("max hop", hp.max_hop)
