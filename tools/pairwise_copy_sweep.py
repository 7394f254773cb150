#!/usr/bin/env python3
"""configs[2] whole call against the blocks per CU of the feature-copy kernel that runs beside the tile kernel (GPU box)."""
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
emb = torch.randn((n, 128), device=dev)
for blocks in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 8, 32]:
    lib.pope_debug_set(_lib.KNOB_COPY_BATCHES, blocks)
    for _ in range(3):
        engine.pairwise_features(x, emb, anchors, "euclidean")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        engine.pairwise_features(x, emb, anchors, "euclidean")
    e1.record()
    torch.cuda.synchronize()
    print("copy batches per wave", blocks, "whole call %.1f us" % (e0.elapsed_time(e1) * 50))
