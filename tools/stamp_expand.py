#!/usr/bin/env python3
"""Diagnostic (GPU box): where does a wave of k_bfs_expand spend its time?  Uses the -DPOPE_STAMP build.

    make -C graphpope_amd/csrc stamp && gpurun -- python tools/stamp_expand.py [level] [flickr|rmat22|rmat20] [K]

Stamps of a wave: 0 entry; of its LAST chunk: 7 chunk begins, 1 indices + live look-ups in, 2 gathers + row masks in (first tile),
3 scan done (first tile), 4 frontier stored (first tile), 5 every tile stored + live bits marked; 6 exit.
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libgraphpope_hip_stamp.so")
from graphpope_amd import engine, synth  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gname = sys.argv[2] if len(sys.argv) > 2 else "flickr"
k = int(sys.argv[3]) if len(sys.argv) > 3 else 256
lib = _lib.load()
dev = engine.require_gpu()
if gname == "flickr":
    ei_np, n = synth.flickr_like()
else:
    path = f"/tmp/{gname}.npy"
    if os.path.exists(path):
        ei_np, n = np.load(path), 1 << int(gname[4:])
    else:
        ei_np, n = synth.rmat(int(gname[4:]), edge_factor=8, seed=1)
anchors = synth.seeded_anchors(n, k, 42)
ei = torch.as_tensor(ei_np, device=dev)
csr = engine.build_csr(ei, n)
for _ in range(3):
    engine.bfs(csr, anchors)
torch.cuda.synchronize()
lib.pope_debug_set_stamp_level.argtypes = [ctypes.c_int]
lib.pope_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pope_debug_set_stamp_level(level)
engine.bfs(csr, anchors)
torch.cuda.synchronize()
buf = np.zeros(16384 * 8, dtype=np.uint64)
lib.pope_debug_read_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(16384, 8).astype(np.int64)
nchunks = (ei_np.shape[1] + 255) // 256
work = st[:min(nchunks, 16384)]          # waves that had a chunk (wave id < nchunks)
work = work[work[:, 0] > 0]
t0 = work[:, 0].min()
print(f"{gname} K={k} level {level}: {nchunks} chunks, {len(work)} stamped working waves ({nchunks / max(len(work), 1):.1f} chunks per wave); ticks are 10 ns")
order = [0, 7, 1, 2, 3, 4, 5, 6]
names = {0: "entry", 7: "last chunk begins", 1: "idx + live in", 2: "gathers + masks in", 3: "scan done", 4: "tile 0 stored", 5: "all tiles + live bits", 6: "exit"}
for i in order:
    col = (work[:, i] - t0) / 100.0
    print(f"  {names[i]:22s} median {np.median(col):8.2f} us   p10 {np.percentile(col, 10):8.2f}   p90 {np.percentile(col, 90):8.2f}   max {col.max():8.2f}")
seq = work[:, order]
d = np.diff(seq, axis=1) / 100.0
print("  phase durations of the last chunk (median us):", {names[order[i + 1]]: round(float(np.median(d[:, i])), 2) for i in range(1, 6)})
print(f"  per chunk on average: {float(np.median((work[:, 6] - work[:, 0]) / 100.0)) / max(nchunks / max(len(work), 1), 1):.2f} us (exit - entry over the chunks of a wave)")
