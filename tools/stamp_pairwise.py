#!/usr/bin/env python3
"""Where does a tile of k_pairwise_persistent go?  (GPU box)   make -C graphpope_amd/csrc stamp && python tools/stamp_pairwise.py [knob] [F]
Stamps (shader clock) of every block's sixth tile.
  consumer waves 0-3: 0 top of tile | 1 last MFMA issued | 2 after the barrier | 3 epilogue of the first 32 columns done | 4 of the second
  loader waves 4-7:   0 top (after the barrier) | 1 DMAs issued | 2 copy issued | 3 everything landed | 4 after the barrier"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libgraphpope_hip_stamp.so")
from graphpope_amd import engine, synth
lib = _lib.load(); dev = engine.require_gpu()
knob = int(sys.argv[1]) if len(sys.argv) > 1 else 0
F = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib.pope_debug_set(_lib.KNOB_PAIRWISE_KERNEL, knob)
n = synth.FLICKR_N
anchors = synth.seeded_anchors(n, 256, 42)
emb = torch.randn((n, 128), device=dev)
if len(sys.argv) > 3 and sys.argv[3] == "ones":
    emb = torch.ones((n, 128), device=dev); emb[:, 0] = torch.arange(n, device=dev) % 7
if len(sys.argv) > 3 and sys.argv[3] == "rand":
    emb = torch.rand((n, 128), device=dev)
x = torch.rand((n, F), device=dev)
for _ in range(3):
    engine.pairwise_features(x, emb, anchors, "euclidean")
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
lib.pope_debug_read_pairwise_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pope_debug_read_pairwise_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(256, 8, 8).astype(np.int64)
print("knob", knob, "F", F)
t0 = st[:, :, 0].min(axis=1, keepdims=True)
for w in range(8):
    rel = st[:, w, :5] - t0
    print(f"wave {w}: median stamps relative to the block's earliest wave: " + "  ".join(f"{np.median(rel[:, i]):7.0f}" for i in range(5)))

e, b0, end = st[:, 0, 6], st[:, 0, 7], st[:, 0, 5]
print("entry -> past B_0: median %.0f cycles (min %.0f, max %.0f); B_0 -> all steps done: median %.0f; entry spread over blocks: %.0f; last end - first entry: %.0f"
      % (np.median(b0 - e), (b0 - e).min(), (b0 - e).max(), np.median(end - b0), e.max() - e.min(), end.max() - e.min()))
