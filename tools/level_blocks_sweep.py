#!/usr/bin/env python3
"""configs[1] step against the cap on the level kernel's expand blocks (GPU box): fewer blocks = fewer copies of the live
table into LDS and a cheaper empty launch, but fewer waves to hide the gather latency of the dense levels."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth
import bench
dev = engine.require_gpu()
lib = _lib.load()
ei_np, n = synth.flickr_like()
anchors = synth.seeded_anchors(n, 256, 42)
x = torch.rand((n, 500), device=dev)
ei = torch.as_tensor(ei_np, device=dev)
for rnd in range(2):
    for blocks in [int(v) for v in sys.argv[1:]] or [0, 660, 440, 293, 220]:
        lib.pope_debug_set(_lib.KNOB_LEVEL_BLOCKS, blocks)
        for _ in range(5):
            bench.pope_step(x, ei, n, anchors, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            bench.pope_step(x, ei, n, anchors, 1)
        torch.cuda.synchronize()
        print("expand blocks cap", blocks, "step %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
