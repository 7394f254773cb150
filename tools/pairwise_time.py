#!/usr/bin/env python3
"""configs[2] timing (GPU box): whole pairwise call (features + embedding) and the embedding alone, per kernel with --trace."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from graphpope_amd import engine, synth
dev = engine.require_gpu()
n = synth.FLICKR_N
anchors = synth.seeded_anchors(n, 256, 42)
x = torch.rand((n, 500), device=dev)
r = bench.pairwise_leg(n, anchors, x, dev, steps=20)
print(json.dumps({k: r[k] for k in ("ms_per_call", "embedding_only_ms", "whole_call_frac_of_mfma_peak", "max_abs_err_vs_sklearn")}))
print(json.dumps(r["roofline"]))
