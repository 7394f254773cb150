#!/usr/bin/env python3
"""A/B of the configs[1] step over POPE_KNOB_FINALIZE_VARIANT (x POPE_KNOB_FINALIZE_BLOCKS): 1 = the pipelined kernel, rows dealt round-robin
(default since round 4), 5 = pipelined with contiguous row blocks, 7 = the round 1-3 kernel, 3 / 4 = embedding columns and feature copy as
two launches.  Interleaved, every setting checked bit-exact against the round 1-3 kernel on a poisoned buffer (GPU box)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth
lib = _lib.load()
dev = engine.require_gpu()
ei_np, n = synth.flickr_like(seed=1)
x = torch.rand((n, 500), device=dev)
ei = torch.as_tensor(ei_np, device=dev)
anchors = synth.seeded_anchors(n, 256, 42)


def run(v, steps=40):
    v, blocks = (v if isinstance(v, tuple) else (v, 0))
    lib.pope_debug_set(_lib.KNOB_FINALIZE_BLOCKS, blocks)
    lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, v)
    for _ in range(3):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


ref = run(7, 5)[1].clone()
times = {v: [] for v in (7, 1)}
for rnd in range(5):
    for v in times:
        ms, out = run(v)
        times[v].append(ms)
        if rnd == 0:
            out.fill_(float("nan")); del out
            out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
            torch.cuda.synchronize()
            if not torch.equal(out, ref): print("MISMATCH", v, flush=True)
            del out
lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 1)
lib.pope_debug_set(_lib.KNOB_FINALIZE_BLOCKS, 0)
for v, t in times.items():
    print(f"finalize variant {v}: median {np.median(t):.4f} ms  min {min(t):.4f}  all {' '.join('%.4f' % a for a in t)}", flush=True)
