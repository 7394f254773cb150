#!/usr/bin/env python3
"""Per-level times of the BFS on the bench graph under every live-table mode (POPE_KNOB_LIVE_MODE: 1 table staged in LDS by every block,
2 table read from global memory, 3 global table behind a summary in LDS + a summary launch per level): what would a per-level choice
between them be worth?  (GPU box)   python3 tools/live_mode_levels.py [K]"""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, _lib
dev = engine.require_gpu()
lib = _lib.load()
ei_np, n = synth.flickr_like()
ei = torch.as_tensor(ei_np, device=dev)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 256
anchors = synth.seeded_anchors(n, k, 42)
csr = engine.build_csr(ei, n)
res = {}
for rnd in range(3):
    for mode in (1, 2, 3):
        lib.pope_debug_set(_lib.KNOB_LIVE_MODE, mode)
        for _ in range(3): engine.bfs(csr, anchors)
        torch.cuda.synchronize()
        lib.pope_profile_levels(1)
        for _ in range(30): hp = engine.bfs(csr, anchors)
        torch.cuda.synchronize()
        cap = 4096
        lv = (ctypes.c_int32 * cap)(); ex = (ctypes.c_float * cap)()
        cnt = lib.pope_profile_read(lv, ex, cap)
        lib.pope_profile_levels(0)
        per = {}
        for i in range(cnt): per.setdefault(lv[i], []).append(ex[i])
        res.setdefault(mode, []).append({l: float(np.median(v)) * 1e3 for l, v in sorted(per.items())})
lib.pope_debug_set(_lib.KNOB_LIVE_MODE, -1)
med = {m: {l: round(float(np.median([r[l] for r in runs])), 2) for l in runs[0]} for m, runs in res.items()}
for m in med: print("mode", m, med[m], "sum", round(sum(med[m].values()), 1))
best = {l: min(med[m][l] for m in (1, 2)) for l in med[1]}
print("per-level best of modes 1 and 2:", best, "sum", round(sum(best.values()), 1), "(events between the launches add ~6 us per level to every figure)")
