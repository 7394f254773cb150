#!/usr/bin/env python3
"""Secondary measurements (GPU box): config 3 (node2vec-euclidean pairwise), the PCIe-inclusive drop-in call,
config 5's per-GPU share on a smaller R-MAT (64 anchors, W = 1).  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, utils as gp

dev = engine.require_gpu()
res = {}

def timed(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

# config 3: Flickr node2vec-euclidean 256 anchors
n, d, k = synth.FLICKR_N, 128, 256
emb = torch.randn(n, d, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
x = torch.rand(n, 500, device=dev)
anchors = synth.seeded_anchors(n, k, 42)
for fn in ("euclidean", "distance", "similarity"):
    t = timed(lambda: engine.pairwise_features(x, emb, anchors, fn))
    res[f"pairwise_{fn}"] = {"ms": t * 1e3, "embeddings_per_s": n * k / t, "tflops_dot": 2.0 * n * k * d / t / 1e12}

# PCIe-inclusive drop-in call: host tensors in, host tensor out
ei_np, n = synth.flickr_like()
class D: pass
data = D(); data.x = torch.rand(n, 500); data.edge_index = torch.as_tensor(ei_np); data.num_nodes = n
def call():
    gp.clear_cache(); np.random.seed(42)
    return gp.Graphpope(data, "flickr", "geodesic", "stochastic", 256, None, 6)
import contextlib, io
with contextlib.redirect_stdout(io.StringIO()):
    t = timed(call, reps=5, warm=2)
res["graphpope_host_to_host"] = {"ms": t * 1e3, "embeddings_per_s": n * 256 / t}

# config 5 per-GPU share on R-MAT scale 20 (1 M nodes, ~31 M CSR slots), 64 anchors
t0 = time.perf_counter(); ei, nn = synth.rmat(20, edge_factor=8, seed=1); gen = time.perf_counter() - t0
eid = torch.as_tensor(ei, device=dev)
anc = synth.seeded_anchors(nn, 64, 42)
t = timed(lambda: engine.geodesic_run(None, eid, nn, anc, want_out=False, reuse_workspace=True), reps=5, warm=2)
_, hp = engine.geodesic_run(None, eid, nn, anc, want_out=False)
res["rmat20_64anchors_bfs_only"] = {"ms": t * 1e3, "N": nn, "E": int(ei.shape[1]), "max_hop": hp.max_hop,
                                    "embeddings_per_s": nn * 64 / t, "per_source_model_gbs": 64 * (4.0 * ei.shape[1] + 8.0 * nn) / t / 1e9,
                                    "graph_generation_s": gen}
# anchor-count sweep on the Flickr-shaped graph (config 4's per-GPU share is 128 anchors; 1 024 = all of config 4 on one GPU)
xf = torch.rand(synth.FLICKR_N, 500, device=dev)
ei_s, n_s = synth.flickr_like()
eis = torch.as_tensor(ei_s, device=dev)
res["anchor_sweep"] = {}
for kk in (64, 128, 256, 512, 1024):
    anc = synth.seeded_anchors(n_s, kk, 42)
    t = timed(lambda: engine.geodesic_run(xf, eis, n_s, anc, reuse_workspace=True), reps=20, warm=3)
    res["anchor_sweep"][str(kk)] = {"ms": t * 1e3, "embeddings_per_s": n_s * kk / t}

# closeness-centrality anchors (utils.py:50-54): every node an anchor, 256 at a time, on the Flickr-shaped graph
ei_f, n_f = synth.flickr_like()
eif = torch.as_tensor(ei_f, device=dev)
engine.closeness_centrality(eif, n_f)
torch.cuda.synchronize(); t0 = time.perf_counter()
score = engine.closeness_centrality(eif, n_f)
res["closeness_centrality_flickr"] = {"s": time.perf_counter() - t0, "batches": -(-n_f // 256), "bfs_sources_per_s": n_f / (time.perf_counter() - t0)}
print(json.dumps(res))
