#!/usr/bin/env python3
"""BASELINE config 5, one GPU's share (GPU box): R-MAT scale 22 (4.2 M nodes, ~64 M CSR slots), 64 of the 512 anchors.
BFS-only (F = 0 in that config) + finalise into a [N, 64] matrix; bit-exact check against the C oracle."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth
dev = engine.require_gpu()
t0 = time.perf_counter(); ei, n = synth.rmat(22, edge_factor=8, seed=1); gen = time.perf_counter() - t0
print("generated", ei.shape, f"{gen:.1f}s", flush=True)
anchors = synth.seeded_anchors(n, 512, 42)[:64]
eid = torch.as_tensor(ei, device=dev)
x0 = torch.zeros((n, 0), device=dev)
def run(): return engine.geodesic_run(x0, eid, n, anchors, reuse_workspace=True)
for _ in range(2): out, hp = run()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): out, hp = run()
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
res = {"N": n, "E": int(ei.shape[1]), "K_gpu": 64, "ms": t * 1e3, "embeddings_per_s": n * 64 / t, "max_hop": hp.max_hop,
       "per_source_model_GBps": 64 * (4.0 * ei.shape[1] + 8.0 * n) / t / 1e9, "gen_s": gen}
print(json.dumps(res), flush=True)
if "--check" in sys.argv:
    from oracle import oracle
    t0 = time.perf_counter(); want = oracle.geodesic_hops(ei, n, anchors); cpu = time.perf_counter() - t0
    got = engine.hop_matrix(hp).cpu().numpy()
    print(json.dumps({"bit_exact": bool(np.array_equal(got, want)), "oracle_s": cpu, "oracle_embeddings_per_s": n * 64 / cpu,
                      "emb_exact": bool(np.array_equal(out.cpu().numpy().view(np.uint32), oracle.hops_to_embedding(want).view(np.uint32)))}))
if "--all512" in sys.argv:
    # all of config 5 on ONE GPU: 512 anchors (W = 8 words, two 4-word tiles per node); 8 random columns checked against the oracle
    anchors512 = synth.seeded_anchors(n, 512, 42)
    def run512(): return engine.geodesic_run(x0, eid, n, anchors512, reuse_workspace=True)
    for _ in range(2): out5, hp5 = run512()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out5, hp5 = run512()
    torch.cuda.synchronize(); t5 = (time.perf_counter() - t0) / 3
    r5 = {"K": 512, "ms": t5 * 1e3, "embeddings_per_s": n * 512 / t5, "max_hop": hp5.max_hop}
    if "--check" in sys.argv:
        from oracle import oracle
        cols = np.random.RandomState(0).choice(512, 8, replace=False)
        want = oracle.geodesic_hops(ei, n, anchors512[cols])
        got = engine.hop_matrix(hp5)[:, torch.as_tensor(cols, device=dev)].cpu().numpy()
        r5["sampled_columns_bit_exact"] = bool(np.array_equal(got, want))
    print(json.dumps(r5), flush=True)

