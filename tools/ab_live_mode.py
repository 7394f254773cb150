#!/usr/bin/env python3
"""A/B of the live-table modes of k_bfs_level on graphs too large for the LDS table (GPU box).
mode 0 = no table, 2 = table read from global memory.  usage: python tools/ab_live_mode.py [scale] [edge_factor]"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, _lib
dev = engine.require_gpu()
lib = _lib.load()
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ei, n = synth.rmat(scale, edge_factor=ef, seed=1)
eid = torch.as_tensor(ei, device=dev)
res = {"N": n, "E": int(ei.shape[1])}
ref = None
for k in (64, 256):
    anc = synth.seeded_anchors(n, k, 42)
    for mode in (0, 2, 0, 2):
        lib.pope_debug_set(_lib.KNOB_LIVE_MODE, mode)
        for _ in range(2): engine.geodesic_run(None, eid, n, anc, want_out=False, reuse_workspace=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): _, hp = engine.geodesic_run(None, eid, n, anc, want_out=False, reuse_workspace=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        h = engine.hop_matrix(hp)
        if mode == 0 and ref is None or (ref is not None and ref.shape != h.shape): ref = h.clone()
        assert torch.equal(h, ref), "modes disagree"
        res.setdefault(f"k{k}_mode{mode}_ms", []).append(round(dt * 1e3, 3))
    ref = None
    res[f"k{k}_max_hop"] = hp.max_hop
lib.pope_debug_set(_lib.KNOB_LIVE_MODE, -1)
print(json.dumps(res))
