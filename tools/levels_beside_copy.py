#!/usr/bin/env python3
"""How long does each BFS level launch take while a feature copy saturates the memory system on ANOTHER stream (GPU box)?
The concept probe for running the sparse levels underneath out[:, :F] = x: per-level medians alone / beside the copy."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, _lib

dev = engine.require_gpu()
lib = _lib.load()
ei_np, n = synth.flickr_like()
ei = torch.as_tensor(ei_np, device=dev)
anchors = synth.seeded_anchors(n, 256, 42)
csr = engine.build_csr(ei, n)
F = 500
x = torch.rand((n, F), device=dev)
out = torch.empty((n, F + 256), device=dev)
side = torch.cuda.Stream()


hp0 = engine.bfs(csr, anchors)
planes0 = hp0.planes.clone()
bits0 = hp0.n_hop_bits


def levels(beside, copies=4, blocks=0):
    lib.pope_debug_set(_lib.KNOB_FINALIZE_BLOCKS, blocks)
    for _ in range(3): engine.bfs(csr, anchors)
    torch.cuda.synchronize()
    lib.pope_profile_levels(1)
    for _ in range(10):
        if beside:
            with torch.cuda.stream(side):
                for _ in range(copies):
                    if blocks: engine.finalize(planes0, bits0, n, 256, x, F, out)
                    else: engine.copy_features(x, F, out)
        engine.bfs(csr, anchors)
        torch.cuda.synchronize()
    capn = 4096
    lv = (ctypes.c_int32 * capn)(); ex = (ctypes.c_float * capn)()
    cnt = lib.pope_profile_read(lv, ex, capn)
    lib.pope_profile_levels(0)
    per = {}
    for i in range(cnt): per.setdefault(lv[i], []).append(ex[i])
    return {l: round(1e3 * float(np.median(v)), 1) for l, v in sorted(per.items())}


print("alone       ", json.dumps(levels(False)), flush=True)
print("beside copy ", json.dumps(levels(True)), flush=True)
for blocks in (1024, 512, 256, 128):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.pope_debug_set(_lib.KNOB_FINALIZE_BLOCKS, blocks)
    engine.finalize(planes0, bits0, n, 256, x, F, out); torch.cuda.synchronize()
    e0.record(); engine.finalize(planes0, bits0, n, 256, x, F, out); e1.record(); torch.cuda.synchronize()
    print(f"beside finalise kernel of {blocks} blocks ({e0.elapsed_time(e1) * 1e3:.0f} us alone)", json.dumps(levels(True, 6, blocks)), flush=True)
lib.pope_debug_set(_lib.KNOB_FINALIZE_BLOCKS, 0)
print("alone again ", json.dumps(levels(False)), flush=True)
