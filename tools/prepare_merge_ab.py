#!/usr/bin/env python3
"""pope_geodesic_run with the clear + seed + CSR build as one launch (POPE_KNOB_PREPARE_MERGE) against the two launches (GPU box):
python tools/prepare_merge_ab.py [steps]   -- configs[1] step, alternating, wall clock over `steps` calls each."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
modes = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1]
lib = _lib.load()
dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
anchors = synth.seeded_anchors(n, 256, 42)
ei = torch.as_tensor(ei_np, device=dev)
x = torch.rand((n, 500), device=dev)
ref = None
for rnd in range(4):
    for merge in modes:
        lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, merge)
        for _ in range(10):
            out, hp = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out, hp = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        if ref is None:
            ref = out.clone()
        print(f"round {rnd}  merge {merge}: {ms:.4f} ms per step   output identical to the first: {torch.equal(out, ref)}", flush=True)
lib.pope_debug_set(_lib.KNOB_PREPARE_MERGE, 1)
