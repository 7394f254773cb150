#!/usr/bin/env python3
"""A/B of tile shape / split-K for the SAGE layer-0 forward (gather excluded) and the twin weight gradient (GPU box)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib
if os.environ.get('POPE_LIB'): _lib.LIB_PATH = os.environ['POPE_LIB']
from graphpope_amd import engine
from graphpope_amd.sage import SAGEConv, SampledAdj
lib = _lib.load(); dev = engine.require_gpu()
n_dst, n_src, c_in, c_out = 9988, 37799, 756, 256
rowptr = torch.zeros(n_dst + 1, dtype=torch.int32)                       # no neighbours: the gather is a few us of zero fill
adj = SampledAdj(rowptr, torch.zeros(0, dtype=torch.int32), n_src).to(dev)
conv = SAGEConv(c_in, c_out).to(dev); x = torch.randn(n_src, c_in, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def fwd():
    with torch.no_grad(): conv((x, x[:n_dst]), adj)
for tile, splits in [(0, 1), (1, 1), (2, 1), (1, 2)]:
    lib.pope_debug_set(_lib.KNOB_GEMM_TILE, tile)
    print(f"forward tile {('auto','64x64','64x128','128x256')[tile]:8s} splits {splits}: {timed(fwd):7.1f} us")
lib.pope_debug_set(_lib.KNOB_GEMM_TILE, 0)
