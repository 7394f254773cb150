#!/usr/bin/env python3
"""A/B of the configs[1] step (pope_geodesic_run, inputs resident) over the tail kernel (k_tail_finalize) (GPU box):
    python tools/tail_ab.py [K] [spec ...]      spec = "T" or "T:B" (first level inside the tail kernel : its BFS blocks), 0 = off
Interleaved rounds in one process, median + min per setting; every setting is checked bit-exact against the first one
on a poisoned output buffer, with and without an output matrix (hop planes compared through pope_geodesic_hops)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth  # noqa: E402

lib = _lib.load()
dev = engine.require_gpu()
args = sys.argv[1:]
K = int(args.pop(0)) if args and args[0].isdigit() and int(args[0]) >= 32 else 256
ei_np, n = synth.flickr_like(seed=1)
F = 500
x = torch.rand((n, F), device=dev)
ei = torch.as_tensor(ei_np, device=dev)
anchors = synth.seeded_anchors(n, K, 42)
specs = args or ["0", "8:256", "8:128", "8:384", "9:256", "7:256", "6:256", "10:256", "11:256", "8:64", "8:512"]


def apply(spec):
    spec, _, variant = spec.partition("/")
    assert lib.pope_debug_set(_lib.KNOB_LEVEL_VARIANT, int(variant) if variant else 0) == 0
    t, _, b = spec.partition(":")
    assert lib.pope_debug_set(_lib.KNOB_TAIL_LEVEL, int(t)) == 0
    assert lib.pope_debug_set(_lib.KNOB_TAIL_BLOCKS, int(b) if b else 256) == 0


WITH_OUT = os.environ.get("TAIL_AB_NO_OUT") is None


def run(spec, steps):
    apply(spec)
    if not WITH_OUT:
        for _ in range(3):
            engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            engine.geodesic_run(None, ei, n, anchors, want_out=False, reuse_workspace=True)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3, engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    for _ in range(3):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


apply(specs[0])
ref = run(specs[0], 5)[1].clone()
ref_hops = engine.hop_matrix(engine.geodesic_run(None, ei, n, anchors, want_out=False)[1]).clone()
times = {s: [] for s in specs}
for rnd in range(5):
    for s in specs:
        ms, out = run(s, 40)
        times[s].append(ms)
        if rnd == 0:
            out.fill_(float("nan"))
            del out
            out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad = (out != ref) | torch.isnan(out)
                print("MISMATCH", s, int(bad.sum()), "elements; first rows", torch.nonzero(bad.any(1))[:5].flatten().tolist(), flush=True)
            del out
            hp = engine.geodesic_run(None, ei, n, anchors, want_out=False)[1]
            if not torch.equal(engine.hop_matrix(hp), ref_hops):
                print("MISMATCH (planes only)", s, flush=True)
apply("0")
for s in specs:
    t = times[s]
    print(f"{s:12s} median {np.median(t):.4f} ms  min {min(t):.4f}  all {' '.join('%.4f' % v for v in t)}", flush=True)
