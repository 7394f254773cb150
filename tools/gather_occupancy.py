#!/usr/bin/env python3
"""k_gather_mean of the config-2 layer-0 block at capped occupancy (GPU box): python tools/gather_occupancy.py
Question behind it: could the gather run beside the forward GEMM inside one launch, where every block carries the GEMM's
LDS reservation?  Times the indexed gather (feats[n_id], x_dst copy) alone, warm and after a cache flush."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth
from graphpope_amd import sage as gs
lib = _lib.load(); dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
feats = torch.rand(n, 756, device=dev)
from graphpope_amd.sampler import NeighborSampler
csr = engine.build_csr(torch.as_tensor(ei_np).to(dev), n)
smp = NeighborSampler(csr.rowptr, csr.col, n, (25, 10))
torch.manual_seed(3)
seeds = torch.randperm(n, device=dev)[:1550]
n_id, adjs = smp.sample(seeds, seed=1)
adj = adjs[0]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
conv = gs.SAGEConv(756, 256).to(dev)
x = gs.IndexedFeatures(feats, n_id)
print("block", adj.n_dst, adj.n_src, int(adj.rowptr[-1]))
MODES = [(1, 0), (0, 0), (1, 0), (0, 0)] if "--overlap" in sys.argv else [(0, p) for p in (0, 20, 40, 53, 80, 0)]
for overlap, pad in MODES:
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, overlap)
    lib.pope_debug_set(_lib.KNOB_GATHER_LDS_PAD_KB, pad)
    for cold in (False, True):
        ts = []
        with torch.no_grad():
            for _ in range(8):
                if cold:
                    flush.fill_(1)
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                out = conv(x, adj)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"overlap {overlap} lds pad {pad:3d} KB  {'after a 512 MB flush' if cold else 'warm':22s} layer-0 forward (gather + projection) median {np.median(ts[2:]):7.1f} us  min {min(ts[2:]):7.1f}")
lib.pope_debug_set(_lib.KNOB_GATHER_LDS_PAD_KB, 0)
lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
