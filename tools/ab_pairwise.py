#!/usr/bin/env python3
"""Diagnostic builds of k_pairwise (PW_VARIANT): where does its time go?  (GPU box)"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
from graphpope_amd import _lib
if sys.argv[1] != "default": _lib.LIB_PATH = sys.argv[1]
import torch, numpy as np, ctypes
from graphpope_amd import engine, synth
from graphpope_amd._lib import ptr, check
dev = engine.require_gpu(); lib = _lib.load()
n, d, k = synth.FLICKR_N, 128, 256
emb = torch.randn(n, d, device=dev); out = torch.empty(n, 756, device=dev)
a = emb[torch.as_tensor(synth.seeded_anchors(n, k, 42), device=dev)].contiguous()
scratch = torch.empty(lib.pope_pairwise_scratch_bytes(n, k, d), dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(): check(lib.pope_pairwise_minmax(ptr(emb), n, d, ptr(a), k, 2, ptr(out), 756, 500, ptr(scratch), scratch.numel(), st))
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"us_per_call": e0.elapsed_time(e1) * 1e3 / 20}))
''' % ROOT
for lib in ["default"] + sys.argv[1:]:
    out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
    print(os.path.basename(lib), out.stdout.strip().split("\n")[-1] if out.returncode == 0 else out.stderr[-400:])
