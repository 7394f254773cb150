#!/usr/bin/env python3
"""Per-level k_bfs_level durations under caps on the expand blocks (POPE_KNOB_LEVEL_BLOCKS), Flickr-shaped graph (GPU box).
Which levels (sparse / dense) gain or lose from fewer, longer-lived blocks."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, _lib

dev = engine.require_gpu()
lib = _lib.load()
ei_np, n = synth.flickr_like()
ei = torch.as_tensor(ei_np, device=dev)
anchors = synth.seeded_anchors(n, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 42)
csr = engine.build_csr(ei, n)
for cap in (0, 660, 440, 293, 220, 128, 64):
    lib.pope_debug_set(_lib.KNOB_LEVEL_BLOCKS, cap)
    for _ in range(3): hp = engine.bfs(csr, anchors)
    torch.cuda.synchronize()
    lib.pope_profile_levels(1)
    for _ in range(20): hp = engine.bfs(csr, anchors)
    torch.cuda.synchronize()
    capn = 4096
    lv = (ctypes.c_int32 * capn)(); ex = (ctypes.c_float * capn)()
    cnt = lib.pope_profile_read(lv, ex, capn)
    lib.pope_profile_levels(0)
    per = {}
    for i in range(cnt): per.setdefault(lv[i], []).append(ex[i])
    print(cap, json.dumps({l: round(1e3 * float(np.median(v)), 1) for l, v in sorted(per.items())}), flush=True)
lib.pope_debug_set(_lib.KNOB_LEVEL_BLOCKS, 0)
