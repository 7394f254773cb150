#!/usr/bin/env python3
"""The large-graph step (BASELINE configs[4]: R-MAT scale 22) or the configs[3] per-rank shapes, as a plain loop a profiler can wrap.

    python3 tools/big_graph_run.py <graph> <K> [steps] [mode]
        graph: rmat22 | rmat20 | flickr        K: anchors of this call
        mode:  run (default: engine.geodesic_run, F = 0 for R-MAT, F = 500 for flickr) | levels (per-level HIP event times + live nodes)
             | bfs (BFS only, no output matrix) | shards (what a rank of an 8-GPU run does behind its all-gather: pope_geodesic_finalize_shards
               of 8 shards of K anchors each on made-up planes, beside it the feature copy; HIP events)

The R-MAT edge list takes 13 s to generate: it is cached as /tmp/<graph>.npy for the other processes of the same GPU call.
Prints one JSON line.
"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth  # noqa: E402

if os.environ.get("POPE_LIB"):
    _lib.LIB_PATH = os.environ["POPE_LIB"]


def graph(name):
    if name == "flickr":
        return synth.flickr_like(seed=1)
    scale = int(name[4:])
    path = f"/tmp/{name}.npy"
    if os.path.exists(path):
        return np.load(path), 1 << scale
    ei, n = synth.rmat(scale, edge_factor=8, seed=1)
    try:
        np.save(path + ".tmp.npy", ei)
        os.replace(path + ".tmp.npy", path)
    except OSError:
        pass
    return ei, n


def main():
    name, k = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    mode = sys.argv[4] if len(sys.argv) > 4 else "run"
    dev = engine.require_gpu()
    lib = _lib.load()
    ei_np, n = graph(name)
    e = ei_np.shape[1]
    anchors = synth.seeded_anchors(n, k, 42)
    ei = torch.as_tensor(ei_np, device=dev)
    x = None
    if name == "flickr":
        x = torch.rand((n, 500), device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    res = {"graph": name, "N": n, "E": e, "K": k, "mode": mode, "steps": steps}
    if mode == "shards":
        world, bits = 8, 4
        w = lib.pope_words(k)
        planes = torch.randint(-2**62, 2**62, (world, 1 + bits, n, w), dtype=torch.int64, device=dev)
        f = 0 if x is None else x.shape[1]
        out = torch.empty((n, f + world * k), dtype=torch.float32, device=dev)

        def timed(fn):
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / steps
        res["finalize_shards_ms"] = timed(lambda: engine.finalize_shards(planes, bits, n, k, None, f, out))
        res["kernel"] = ""
        import ctypes as _c
        buf = _c.create_string_buffer(64)
        lib.pope_finalize_kernel_name(n, k, f, 0, world, buf, 64)
        res["kernel"] = buf.value.decode()
        res["columns_GB"] = n * world * k * 4 / 1e9
        res["planes_all_gathered_GB"] = planes.numel() * 8 / 1e9
        res["planes_sent_per_rank_GB"] = (1 + bits) * n * w * 8 / 1e9
        if x is not None:
            res["feature_copy_ms"] = timed(lambda: engine.copy_features(x, f, out))
        print(json.dumps(res))
        return
    if mode == "levels":
        csr = engine.build_csr(ei, n)
        for _ in range(2):
            hp = engine.bfs(csr, anchors)
        torch.cuda.synchronize()
        lib.pope_profile_levels(1)
        for _ in range(steps):
            hp = engine.bfs(csr, anchors)
        torch.cuda.synchronize()
        cap = 4096
        lv = (ctypes.c_int32 * cap)()
        ex = (ctypes.c_float * cap)()
        cnt = lib.pope_profile_read(lv, ex, cap)
        lib.pope_profile_levels(0)
        per = {}
        for i in range(cnt):
            per.setdefault(lv[i], []).append(ex[i])
        res["max_hop"] = hp.max_hop
        res["level_us"] = {l: round(1e3 * float(np.median(v)), 1) for l, v in sorted(per.items())}
        res["bfs_levels_sum_us"] = round(sum(res["level_us"].values()), 1)
        # live nodes per frontier from one sampled column block (the whole hop matrix of 4 M x 512 is 8.6 GB as int32)
        wp = hp.planes.shape[2]
        res["words_per_node"] = wp
        res["dense_level_bytes_model"] = e * (8.0 + 8.0 * wp) + n * 16.0 * wp
    else:
        want_out = mode == "run"
        for _ in range(2):
            out, hp = engine.geodesic_run(x, ei, n, anchors, want_out=want_out, reuse_workspace=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out, hp = engine.geodesic_run(x, ei, n, anchors, want_out=want_out, reuse_workspace=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        res["ms_per_step"] = dt * 1e3
        res["max_hop"] = hp.max_hop
        res["embeddings_per_s"] = n * k / dt
    print(json.dumps(res))


if __name__ == "__main__":
    main()
