#!/usr/bin/env python3
"""A/B of the configs[1] step (pope_geodesic_run, inputs resident) over the experiment knobs (GPU box):
    python tools/step_ab.py            # level variants x copy gates, interleaved rounds in one process, median + min per setting
Also checks every setting bit-exact against the default one."""
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
ei_np, n = synth.flickr_like(seed=1)
F, K = 500, int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.rand((n, F), device=dev)
ei = torch.as_tensor(ei_np, device=dev)
anchors = synth.seeded_anchors(n, K, 42)
settings = [("default", 0, 0)]
for v in (1, 2, 3, 4, 5, 7):
    settings.append((f"variant={v}", v, 0))
for g in (40000, 30030, 40030, 50030, 45025, 60040, 30, 35035):
    settings.append((f"gate={g:05d}", 0, g))
settings.append(("variant=3 gate=40030", 3, 40030))


def run(variant, gate, steps):
    lib.pope_debug_set(_lib.KNOB_LEVEL_VARIANT, variant)
    lib.pope_debug_set(_lib.KNOB_COPY_GATE, gate)
    for _ in range(3):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


ref = run(0, 0, 5)[1].clone()
times = {name: [] for name, _, _ in settings}
for rnd in range(5):
    for name, v, g in settings:
        ms, out = run(v, g, 40)
        times[name].append(ms)
        if rnd == 0 and not torch.equal(out, ref):
            print("MISMATCH", name, flush=True)
lib.pope_debug_set(_lib.KNOB_LEVEL_VARIANT, 0)
lib.pope_debug_set(_lib.KNOB_COPY_GATE, 0)
for name, _, _ in settings:
    t = times[name]
    print(f"{name:26s} median {np.median(t):.4f} ms  min {min(t):.4f}  all {' '.join('%.4f' % v for v in t)}", flush=True)
