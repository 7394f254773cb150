#!/usr/bin/env python3
"""The expansion every rank of the multi-GPU path runs after the all-gather (pope_geodesic_finalize_shards: the K columns of ALL
shards in one pass, features copied separately underneath the exchange), timed on ONE GPU with made-up gathered planes:
POPE_KNOB_FINALIZE_VARIANT 7 (rounds 1-3 kernel) against 1 (pipelined, round 4); plus the feature copy (pope_concat).  GPU box."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine
lib = _lib.load()
dev = engine.require_gpu()
n, f, bits = 89250, 500, 4
x = torch.rand((n, f), device=dev)


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for world, k in ((8, 256), (8, 128), (4, 256), (2, 256), (8, 64), (1, 1024)):
    w = lib.pope_words(k)
    planes = torch.randint(-2**62, 2**62, (world, 1 + bits, n, w), dtype=torch.int64, device=dev)
    out = torch.empty((n, f + world * k), device=dev)
    res = {}
    for variant in (7, 10, 9):
        lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, variant)
        res[variant] = timed(lambda: engine.finalize_shards(planes, bits, n, k, None, f, out))
    lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, 9)
    sweep = {}
    for blocks in (1024, 2048, 4096, 8192, 16384):
        lib.pope_debug_set(_lib.KNOB_FINALIZE_BLOCKS, blocks)
        sweep[blocks] = round(timed(lambda: engine.finalize_shards(planes, bits, n, k, None, f, out)), 1)
    lib.pope_debug_set(_lib.KNOB_FINALIZE_BLOCKS, 0)
    print("   round-4 default kernel by grid size:", sweep)
    cp = timed(lambda: engine.copy_features(x, f, out))
    mb = n * world * k * 4 / 1e6
    print("   shuffle kernel k_finalize_wide %.1f us" % res[10])
    print(f"world {world} x {k} anchors: K columns {mb:.0f} MB: rounds 1-3 kernel {res[7]:.1f} us ({mb / res[7]:.2f} TB/s), round-4 default {res[9]:.1f} us ({mb / res[9]:.2f} TB/s); "
          f"feature copy into the [N, {f + world * k}] matrix {cp:.1f} us", flush=True)
    del planes, out
