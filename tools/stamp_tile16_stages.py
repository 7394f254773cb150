#!/usr/bin/env python3
"""Per-stage timeline of the whole-tile forward GEMM from in-kernel stamps (diagnostic build): who arrives last at each stage's
barrier -- the loader wave (its request has not landed: latency-bound) or the consumer wave (MFMA / LDS-bound)?
    make -C graphpope_amd/csrc stamp && gpurun -- python tools/stamp_tile16_stages.py [buffers]"""
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

bufs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
exps = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
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
lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 0)              # gather, then the whole projection (48 stages) in one plain launch
lib.pope_debug_set(_lib.KNOB_GEMM_TILE16_BUFFERS, bufs)
# (round 4 ran this with experiment switches compiled into the diagnostic build -- loader waves without s_setprio 3, no DMA behind
#  stage 4, no fragment reads behind stage 2: profiles/r04_tile16_stage_experiments.txt; the switches are gone again)
for exp in exps[:1]:
    for _ in range(4):
        _lib.check(lib.sage_conv_forward_indexed(_lib.ptr(a0.rowptr), _lib.ptr(a0.col), _lib.ptr(n_id), a0.n_src, a0.n_dst, a0.col.numel(),
                                                 _lib.ptr(feats), n, c_in, _lib.ptr(w_l), _lib.ptr(b), _lib.ptr(w_r), c_out, _lib.ptr(agg),
                                                 _lib.ptr(x_dst), _lib.ptr(out), _lib.ptr(scratch), scratch.numel(), None, stream))
    torch.cuda.synchronize()
    cnt = 256 * 4 * 64
    host = (ctypes.c_ulonglong * cnt)()
    lib.pope_debug_read_t16_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert lib.pope_debug_read_t16_trace(host, cnt) == 0
    tr = np.frombuffer(host, dtype=np.uint64).reshape(256, 4, 64).astype(np.int64)
    live = tr[:, 1, 0] > 0
    tr = tr[live]
    S = int((tr[0, 1] > 0).sum())
    print(f"{bufs} stage buffers; {tr.shape[0]} blocks, {S} stages with a barrier behind them; times in us (100 MHz counter)")
    t0 = tr[:, 2, 0:1]
    stage = np.diff(tr[:, 2, :S], axis=1) / 100.0                       # consumer: barrier exit to barrier exit
    print(f"  stage length (consumer wave 0, exit to exit): median {np.median(stage):.3f}  p10 {np.percentile(stage, 10):.3f}  p90 {np.percentile(stage, 90):.3f}")
    busy = (tr[:, 1, 1:S] - tr[:, 2, 0:S - 1]) / 100.0                  # barrier exit -> next arrival = the stage's own work
    wait = (tr[:, 2, 1:S] - tr[:, 1, 1:S]) / 100.0                      # arrival -> exit = waiting for the others
    print(f"  consumer: work between barriers median {np.median(busy):.3f}, wait at the barrier median {np.median(wait):.3f}  p90 {np.percentile(wait, 90):.3f}")
    late = (tr[:, 0, 1:S] - tr[:, 1, 1:S]) / 100.0                      # loader arrival minus consumer arrival (> 0: loader last)
    print(f"  loader wave 0 arrives after consumer wave 0 in {100.0 * float((late > 0).mean()):.0f} % of the stages; "
          f"loader - consumer arrival: median {np.median(late):+.3f}  p10 {np.percentile(late, 10):+.3f}  p90 {np.percentile(late, 90):+.3f}")
    per_stage = np.median(stage, axis=0)
    print("  median stage length by stage:", " ".join(f"{v:.2f}" for v in per_stage))
    clk = (tr[:, 3, S - 1] - tr[:, 3, 0]) / ((tr[:, 2, S - 1] - tr[:, 2, 0]) / 100.0)
    print(f"  shader clock over the main loop (s_memtime / s_memrealtime): median {np.median(clk):.0f} MHz  min {clk.min():.0f}  max {clk.max():.0f}")
    total = (tr[:, 2, S - 1] - tr[:, 2, 0]) / 100.0
    print(f"  first to last barrier: median {np.median(total):.1f} us over {S - 1} stages = {np.median(total) / (S - 1):.3f} us per stage; 80 MFMAs of 32 clocks = 1.14 us at 2.25 GHz")
