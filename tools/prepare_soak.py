#!/usr/bin/env python3
"""Soak of the one-launch prepare (k_prepare) against the two launches (GPU box): python tools/prepare_soak.py [calls]
Random graphs (R-MAT scales 8-16, Flickr-shaped), random anchor sets of 1-256 with repeats, workspaces reused across
graphs of equal size and scribbled over now and then; every call's hop planes must equal the separate launches' bit for bit."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
lib = _lib.load()
dev = engine.require_gpu()
rng = np.random.RandomState(7)
graphs = []
for scale in (8, 10, 12, 14, 16):
    ei, n = synth.rmat(scale, edge_factor=8, seed=scale)
    graphs.append((torch.as_tensor(ei, device=dev), n))
ei, n = synth.flickr_like()
graphs.append((torch.as_tensor(ei, device=dev), n))
t0 = time.perf_counter()
bad = 0
for c in range(calls):
    eid, n = graphs[rng.randint(len(graphs))]
    k = int(rng.choice([1, 2, 63, 64, 65, 100, 128, 200, 255, 256]))
    anchors = rng.choice(np.arange(n), k)                     # with repeats
    lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 0)
    want = engine.geodesic_run(None, eid, n, anchors, want_out=False)[1].valid().clone()
    lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 1)
    if c % 50 == 7:
        for ws in engine._WORKSPACE.values():
            ws.copy_(torch.randint(0, 256, ws.shape, dtype=torch.uint8, device=ws.device))
    for rep in range(3):
        got = engine.geodesic_run(None, eid, n, anchors, want_out=False, reuse_workspace=True)[1].valid()
        if not torch.equal(got, want):
            bad += 1
            print(f"MISMATCH call {c} rep {rep} n {n} k {k}", flush=True)
    if c % 500 == 499:
        print(f"{c + 1} anchor sets ({3 * (c + 1)} merged calls), {bad} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 1)
print(f"done: {calls} anchor sets x 3 merged calls, {bad} mismatches")
sys.exit(1 if bad else 0)
