#!/usr/bin/env python3
"""Diagnostic (GPU box): where does a wave of k_bfs_expand spend its time?  Uses the -DPOPE_STAMP build.

    make -C graphpope_amd/csrc stamp && gpurun -- python tools/stamp_expand.py [level]
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
lib = _lib.load()
dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
anchors = synth.seeded_anchors(n, 256, 42)
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
work = st[:nchunks]                      # waves that had a chunk (wave id < nchunks)
t0 = st[:8192, 0][st[:8192, 0] > 0].min()
print(f"level {level}: {nchunks} working waves; ticks are 10 ns")
names = ["entry", "idx loaded", "gathers done", "scan done", "frontier stored", "committed", "exit"]
for i, nm in enumerate(names):
    col = (work[:, i] - t0) / 100.0
    print(f"  {nm:16s} median {np.median(col):7.2f} us   p10 {np.percentile(col, 10):7.2f}   p90 {np.percentile(col, 90):7.2f}   max {col.max():7.2f}")
d = np.diff(work[:, :7], axis=1) / 100.0
print("  phase durations (median us):", {names[i + 1]: round(float(np.median(d[:, i])), 2) for i in range(6)})
idle = st[nchunks:8192]
idle = idle[idle[:, 0] > 0]
if len(idle):
    print(f"  idle waves: {len(idle)}, entry median {(np.median(idle[:, 0]) - t0) / 100:.2f} us, exit max {(idle[:, 6].max() - t0) / 100:.2f} us")
