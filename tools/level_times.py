#!/usr/bin/env python3
"""Per-level durations of k_bfs_level on the bench graph (GPU box) + how many nodes are live at each level."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, _lib

if os.environ.get("POPE_LIB"): _lib.LIB_PATH = os.environ["POPE_LIB"]
dev = engine.require_gpu()
lib = _lib.load()
ei_np, n = synth.flickr_like()
ei = torch.as_tensor(ei_np, device=dev)
anchors = synth.seeded_anchors(n, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 42)
csr = engine.build_csr(ei, n)
for _ in range(3): hp = engine.bfs(csr, anchors)
torch.cuda.synchronize()
lib.pope_profile_levels(1)  # per-level events
reps = 20
for _ in range(reps): hp = engine.bfs(csr, anchors)
torch.cuda.synchronize()
cap = 4096
lv = (ctypes.c_int32 * cap)(); ex = (ctypes.c_float * cap)()
cnt = lib.pope_profile_read(lv, ex, cap)
lib.pope_profile_levels(0)
per = {}
for i in range(cnt): per.setdefault(lv[i], []).append(ex[i])
hops = engine.hop_matrix(hp).cpu().numpy()            # [N, K], -1 unreachable
live = {l: int(((hops == l).any(axis=1)).sum()) for l in range(0, hp.max_hop + 1)}
print(json.dumps({"max_hop": hp.max_hop, "level_us": {l: round(1e3 * float(np.mean(v)), 2) for l, v in sorted(per.items())},
                  "live_nodes_in_frontier_of_level": live, "N": n}))
