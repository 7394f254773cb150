#!/usr/bin/env python3
"""A/B of the configs[1] step (pope_geodesic_run, inputs resident) over the copy role of the level launches (GPU box):
    python tools/level_copy_ab.py [K] [spec ...]      spec = "l:permille,l:permille,..." (l = 0: every launch), "off" = none
Interleaved rounds in one process, median + min per setting, every setting checked bit-exact against the first one."""
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
K = int(args.pop(0)) if args and args[0].isdigit() else 256
ei_np, n = synth.flickr_like(seed=1)
F = 500
x = torch.rand((n, F), device=dev)
ei = torch.as_tensor(ei_np, device=dev)
anchors = synth.seeded_anchors(n, K, 42)
DEFAULT = [
    "off",
    "0:91", "0:50", "0:25",
    "1:150,2:150,8:150,9:150,10:150,11:150", "1:100,2:100,8:100,9:100,10:100,11:100",
    "1:200,2:200,8:100,9:100,10:100,11:100", "1:100,2:100", "8:100,9:100,10:100,11:100",
    "3:50,4:50,5:50,6:50,7:50", "3:100,4:100,5:100,6:100,7:100",
    "1:120,2:120,3:40,4:40,5:40,6:40,7:40,8:120,9:120,10:120,11:120",
    "1:50", "1:100", "1:200", "1:300", "4:50", "4:100", "4:200", "9:50", "9:100", "9:200", "9:300",
]
specs = args or DEFAULT


def apply(spec):
    lib.pope_debug_set(_lib.KNOB_LEVEL_COPY, 0)
    if spec in ("off", "default"):
        return
    for item in spec.split(","):
        l, pm = item.split(":")
        assert lib.pope_debug_set(_lib.KNOB_LEVEL_COPY, (int(l) << 16) | int(pm)) == 0


def run(spec, steps):
    apply(spec)
    for _ in range(3):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


ref = run(specs[0], 5)[1].clone()
times = {s: [] for s in specs}
for rnd in range(5):
    for s in specs:
        ms, out = run(s, 40)
        times[s].append(ms)
        if rnd == 0:
            # torch hands the block of a freed result to the next one of the same size: poison it, so that rows nobody copied show
            out.fill_(float("nan"))
            del out
            out = engine.geodesic_run(x, ei, n, anchors, reuse_workspace=True)[0]
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                print("MISMATCH", s, flush=True)
            del out
apply("off")
for s in specs:
    t = times[s]
    print(f"{s:70s} median {np.median(t):.4f} ms  min {min(t):.4f}  all {' '.join('%.4f' % v for v in t)}", flush=True)
