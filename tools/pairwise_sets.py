#!/usr/bin/env python3
"""configs[2]: whole call and embedding alone with one and with two consumer sets in the persistent tile kernel (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth
dev = engine.require_gpu()
lib = _lib.load()
n = synth.FLICKR_N
anchors = torch.as_tensor(synth.seeded_anchors(n, 256, 42).astype(np.int64), device=dev)
x = torch.rand((n, 500), device=dev)
x0 = torch.empty((n, 0), device=dev)
emb = torch.randn((n, 128), device=dev)
for rnd in range(2):
    for knob in (2, 3, 1):
        lib.pope_debug_set(_lib.KNOB_PAIRWISE_KERNEL, knob)
        for name, xs in (("whole call", x), ("embedding alone", x0)):
            for _ in range(3):
                engine.pairwise_features(xs, emb, anchors, "euclidean")
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                engine.pairwise_features(xs, emb, anchors, "euclidean")
            e1.record()
            torch.cuda.synchronize()
            print({2: "one set ", 3: "two sets", 1: "round-1 kernel"}[knob], name, "%.1f us" % (e0.elapsed_time(e1) * 50))
