#!/usr/bin/env python3
"""A/B of the finalise kernel variants in one process, interleaved rounds (GPU box)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth
lib = _lib.load()
dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
anchors = synth.seeded_anchors(n, 256, 42)
x = torch.rand((n, 500), device=dev)
csr = engine.build_csr(torch.as_tensor(ei_np, device=dev), n)
hp = engine.bfs(csr, anchors)
out = torch.empty((n, 756), device=dev)
ref = None
evict = torch.empty(512 * 1024 * 1024 // 4, device=dev)            # > Infinity Cache: every timed launch starts HBM-cold
cases = [(0, 2048), (1, 1024), (1, 2048), (1, 4096), (1, 8192), (2, 2048)]
res = {c: [] for c in cases}
for rnd in range(10):
    for var, blocks in cases:
        lib.pope_debug_set(_lib.KNOB_FINALIZE_VARIANT, var)
        lib.pope_debug_set(_lib.KNOB_FINALIZE_BLOCKS, blocks)
        evict.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        engine.finalize(hp.planes, hp.n_hop_bits, n, 256, x, 500, out, 0)
        e1.record(); torch.cuda.synchronize()
        res[(var, blocks)].append(e0.elapsed_time(e1) * 1e3)
        if ref is None: ref = out.clone()
        assert torch.equal(out, ref), var
byt = 4.0 * n * 500 + 4.0 * n * 756 + 8.0 * n * 4 * (1 + hp.n_hop_bits)
for var in res:
    med = float(np.median(res[var][2:]))
    print(f"variant/blocks {var}: median {med:.1f} us  min {min(res[var]):.1f} us  -> {byt / med / 1e3:.0f} GB/s (event-bracketed, HBM-cold)")
