#!/usr/bin/env python3
"""Phase times of the host -> host Graphpope call (GPU box): python tools/boundary_trace.py [calls]
Python-side phases by wall clock, the assembly's inner phases from pope_debug_boundary_trace."""
import contextlib
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine, synth  # noqa: E402
from graphpope_amd import utils as gp  # noqa: E402

lib = _lib.load()
dev = engine.require_gpu()
if "--after-omp" in sys.argv:                       # what bench.py's CPU legs leave behind: torch's intra-op thread pool, warmed up
    a = torch.randn(3000, 3000)
    for _ in range(5):
        (a @ a).sum()
    print("torch threads", torch.get_num_threads(), flush=True)
    sys.argv.remove("--after-omp")
ei_np, n = synth.flickr_like()
F, K = 500, 256
x = torch.rand(n, F)
ei_cpu = torch.as_tensor(ei_np)
anchors = synth.seeded_anchors(n, K, 42)
names = ["madvise", "wait_chunks/slots", "register/ring_alloc", "enqueue_dma", "join", "stream/event_sync", "unregister", "total"]
trace = (ctypes.c_double * 8)()


POOL = True                                          # the result's pages come from engine's one-entry pool (no faults after call 1)


def one(threads, chunks, label):
    t0 = time.perf_counter()
    with engine.staged(ei_cpu, dev) as ei_dev:
        t1 = time.perf_counter()
        emb = engine.geodesic_features(None, ei_dev, n, anchors, shard=False)
        t2 = time.perf_counter()
    t3 = time.perf_counter()
    out = engine.host_result_tensor(n, F + K) if POOL else torch.empty((n, F + K), dtype=torch.float32)
    t4 = time.perf_counter()
    engine.assemble_host_result(x, emb, out, F, threads=threads, chunks=chunks)
    t5 = time.perf_counter()
    lib.pope_debug_boundary_trace(trace)
    ms = lambda a, b: (b - a) * 1e3
    print(f"{label}: total {ms(t0, t5):6.2f} | stage+enqueue H2D {ms(t0, t1):5.2f} gpu(until verdict) {ms(t1, t2):5.2f} unpin {ms(t2, t3):5.2f} "
          f"alloc {ms(t3, t4):5.2f} assemble {ms(t4, t5):6.2f} :: " + " ".join(f"{nm}={trace[i]:.2f}" for i, nm in enumerate(names)), flush=True)
    del out


for threads, chunks in ((16, 8), (16, 8), (16, 8), (16, 8), (24, 8), (24, 8), (32, 8), (32, 8), (48, 8), (48, 8), (64, 8), (8, 8), (8, 8), (16, 8)):
    one(threads, chunks, f"RING pooled result, threads {threads:2d} chunks {chunks:2d}")
POOL = False
for threads, chunks in ((16, 8), (16, 8), (16, 8), (32, 8), (32, 8)):
    one(threads, chunks, f"RING fresh pages, threads {threads:2d} chunks {chunks:2d}")

lib.pope_debug_set(_lib.KNOB_HOST_RESULT_MODE, 1)
for threads, chunks in ((16, 8), (16, 8), (16, 8), (16, 16), (16, 4), (16, 1), (32, 8), (8, 8)):
    one(threads, chunks, f"REGISTERED threads {threads:2d} chunks {chunks:2d}")


lib.pope_debug_set(_lib.KNOB_FAIL_HOST_REGISTER, 1)           # no registration at all: the runtime's own handling of pageable memory
for threads, chunks in ((16, 8), (16, 8), (32, 8)):
    one(threads, chunks, f"UNREGISTERED threads {threads:2d} chunks {chunks:2d}")
lib.pope_debug_set(_lib.KNOB_FAIL_HOST_REGISTER, 0)
lib.pope_debug_set(_lib.KNOB_HOST_RESULT_MODE, 0)
for threads, chunks in ((16, 8), (16, 8), (16, 8)):
    one(threads, chunks, f"RING again threads {threads:2d} chunks {chunks:2d}")


class Data:
    pass


d = Data()
d.x, d.edge_index, d.num_nodes = x, ei_cpu, n
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    gp.clear_cache()
    np.random.seed(42)
    with contextlib.redirect_stdout(sys.stderr):
        t0 = time.perf_counter()
        out = gp.Graphpope(d, "flickr", "geodesic", "stochastic", K, None, 6)
        dt = (time.perf_counter() - t0) * 1e3
    lib.pope_debug_boundary_trace(trace)
    print(f"Graphpope call {i}: {dt:7.2f} ms :: " + " ".join(f"{nm}={trace[j]:.2f}" for j, nm in enumerate(names)), flush=True)
    del out
