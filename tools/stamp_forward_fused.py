#!/usr/bin/env python3
"""Timeline of the fused layer-0 forward launch from in-kernel stamps (diagnostic build: make -C graphpope_amd/csrc stamp):
make -C graphpope_amd/csrc stamp && gpurun -- python tools/stamp_forward_fused.py"""
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
from graphpope_amd.sage import sample_batch  # noqa: E402

lib = _lib.load()
dev = engine.require_gpu()
ei_np, n = synth.flickr_like(seed=1)
feats = torch.rand((n, 756), device=dev)
rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei_np[0], minlength=n))])
rng = np.random.default_rng(0)
seeds = rng.choice(n, 1550, replace=False)
n_id, adjs = sample_batch(rowptr, ei_np[1], seeds, sizes=(25, 10), rng=rng)
n_id = torch.as_tensor(n_id, device=dev)
a0 = adjs[0].to(dev)
c_in, c_out = 756, 256
g = torch.Generator().manual_seed(5)
w_l, w_r = (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev), (torch.randn(c_out, c_in, generator=g) * 0.05).to(dev)
b = torch.randn(c_out, generator=g).to(dev)
agg = torch.empty((a0.n_dst, c_in), device=dev)
x_dst = torch.empty((a0.n_dst, c_in), device=dev)
out = torch.empty((a0.n_dst, c_out), device=dev)
scratch = torch.empty(max(lib.sage_conv_forward_scratch_bytes(a0.n_dst, c_in, c_out), 16), dtype=torch.uint8, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 2)
for _ in range(4):
    _lib.check(lib.sage_conv_forward_indexed(_lib.ptr(a0.rowptr), _lib.ptr(a0.col), _lib.ptr(n_id), a0.n_src, a0.n_dst, a0.col.numel(),
                                             _lib.ptr(feats), n, c_in, _lib.ptr(w_l), _lib.ptr(b), _lib.ptr(w_r), c_out, _lib.ptr(agg),
                                             _lib.ptr(x_dst), _lib.ptr(out), _lib.ptr(scratch), scratch.numel(), None, stream))
torch.cuda.synchronize()
cnt = 1024 * 8
host = (ctypes.c_ulonglong * cnt)()
lib.pope_debug_read_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.pope_debug_read_gemm_stamps(host, cnt) == 0
st = np.frombuffer(host, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
gemm = st[:256]
gemm = gemm[gemm[:, 0] > 0]
gath = st[256:512]
gath = gath[gath[:, 4] > 0]
t0 = min(gemm[:, 0].min(), gath[:, 4].min())
us = lambda v: (v - t0) / 100.0
print(f"GEMM blocks {len(gemm)}, gather blocks {len(gath)}  (n_dst {a0.n_dst})")
for name, col in (("tile start", 0), ("wait begins", 1), ("wait ends", 2), ("last stage done", 3)):
    v = us(gemm[:, col])
    print(f"  GEMM {name:16s} min {v.min():7.1f}  median {np.median(v):7.1f}  max {v.max():7.1f} us")
w = (gemm[:, 2] - gemm[:, 1]) / 100.0
print(f"  GEMM wait length       min {w.min():7.1f}  median {np.median(w):7.1f}  max {w.max():7.1f} us")
for name, col in (("start", 4), ("end", 5)):
    v = us(gath[:, col])
    print(f"  gather {name:14s} min {v.min():7.1f}  median {np.median(v):7.1f}  max {v.max():7.1f} us")
order = np.argsort(gemm[:, 2])
print("  wait end by block (every 25th, sorted):", " ".join(f"{us(gemm[i, 2]):.0f}" for i in order[::25]))
