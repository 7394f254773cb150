#!/usr/bin/env python3
"""Diagnostic variants of the persistent pairwise kernel: knob 0 normal, 2 no stores, 3 no epilogue, 4 no MFMA; run under
rocprofv3 --kernel-trace and read the k_pairwise_persistent durations in launch order (3 launches per line printed here)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth
dev = engine.require_gpu()
lib = _lib.load()
for n in (8192, 16384, 32768, synth.FLICKR_N):
    anchors = synth.seeded_anchors(n, 256, 42)
    emb = torch.randn((n, 128), device=dev)
    x0 = torch.empty((n, 0), device=dev)
    for knob in (0, 2, 3, 4):
        lib.pope_debug_set(_lib.KNOB_PAIRWISE_KERNEL, knob)
        for _ in range(3):
            engine.pairwise_features(x0, emb, anchors, "euclidean")
        torch.cuda.synchronize()
        print("n", n, "knob", knob)
