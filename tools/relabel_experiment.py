#!/usr/bin/env python3
"""Experiment (round 2): what would a locality-improving node order buy the level kernel?
The graph is relabelled ON THE HOST (cost not counted) three ways and the BFS is timed per level on each:
  identity            the bench graph as generated (ids are a random permutation of the generator's order)
  degree-descending   hubs first: the most-gathered 32-byte frontier rows share 128-byte lines
  bfs-order           nodes in BFS discovery order from the highest-degree node (neighbours get nearby ids)
python tools/relabel_experiment.py   (GPU box)"""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, _lib
dev = engine.require_gpu(); lib = _lib.load()
ei, n = synth.flickr_like()
anchors = synth.seeded_anchors(n, 256, 42)
deg = np.bincount(ei[0], minlength=n)

def bfs_order():
    import scipy.sparse as sp, scipy.sparse.csgraph as cg
    a = sp.csr_matrix((np.ones(ei.shape[1], np.int8), (ei[0], ei[1])), shape=(n, n))
    order = cg.breadth_first_order(a, int(deg.argmax()), directed=False, return_predecessors=False)
    rest = np.setdiff1d(np.arange(n), order)
    return np.concatenate([order, rest])

def relabel(order):                      # order[new_id] = old_id
    new_of_old = np.empty(n, np.int64); new_of_old[order] = np.arange(n)
    e = np.stack([new_of_old[ei[0]], new_of_old[ei[1]]])
    e = e[:, np.lexsort((e[1], e[0]))]
    return e, new_of_old[anchors]

def level_times(e, anc, reps=20):
    csr = engine.build_csr(torch.as_tensor(e, device=dev), n)
    for _ in range(3): hp = engine.bfs(csr, anc)
    torch.cuda.synchronize()
    lib.pope_profile_levels(1)
    for _ in range(reps): hp = engine.bfs(csr, anc)
    torch.cuda.synchronize()
    cap = 4096; lv = (ctypes.c_int32 * cap)(); ex = (ctypes.c_float * cap)()
    cnt = lib.pope_profile_read(lv, ex, cap); lib.pope_profile_levels(0)
    per = {}
    for i in range(cnt): per.setdefault(lv[i], []).append(ex[i])
    lib.pope_profile_levels(2)
    for _ in range(reps): engine.bfs(csr, anc)
    torch.cuda.synchronize()
    cnt = lib.pope_profile_read(lv, ex, cap); lib.pope_profile_levels(0)
    span = sum(ex[i] for i in range(cnt)) / reps * 1e3
    return {l: round(1e3 * float(np.mean(v)), 1) for l, v in sorted(per.items())}, round(span, 1)

res = {}
for name, order in (("identity", np.arange(n)), ("degree-descending", np.argsort(-deg, kind="stable")), ("bfs-order", bfs_order())):
    e, anc = relabel(order)
    per, span = level_times(e, anc)
    res[name] = {"per_level_us_with_6us_event_overhead": per, "all_levels_back_to_back_us": span}
    print(name, json.dumps(res[name]), flush=True)
