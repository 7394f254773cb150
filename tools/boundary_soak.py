#!/usr/bin/env python3
"""A modest soak of the host -> host boundary after round 4's rework (GPU box): 240 Graphpope() calls over three graph sizes and the
two transports (ring / staged), results released and re-allocated in between, ordinary pageable torch transfers
of recycled host buffers interleaved -- every result compared bit for bit with the CPU oracle's; RSS, device memory and file
descriptors printed so that a leak would show.  One pass, no retries."""
import contextlib, gc, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphpope_amd import engine, synth, utils as gp
from oracle import oracle
import psutil

dev = engine.require_gpu()
proc = psutil.Process()
cases = []
for scale, f, k in ((12, 20, 96), (14, 64, 128), (15, 100, 256)):
    ei, n = synth.rmat(scale, edge_factor=8, seed=scale)
    x = np.random.RandomState(scale).rand(n, f).astype(np.float32)
    cases.append((ei, n, x, k))


class Data:
    pass


t0 = time.time()
bad = 0
for it in range(240):
    ei, n, x, k = cases[it % 3]
    mode = ("ring", "staged")[(it // 3) % 2]
    os.environ["GRAPHPOPE_HOST_RESULT"] = mode
    d = Data()
    d.x, d.edge_index, d.num_nodes = torch.as_tensor(x), torch.as_tensor(ei), n
    gp.clear_cache()
    np.random.seed(it)
    with contextlib.redirect_stdout(sys.stderr):
        out = gp.Graphpope(d, "flickr", "geodesic", "stochastic", k, None, 2)
    if it % 8 == 0:                                  # checked against the oracle (the C BFS is the slow part of this script)
        want = oracle.geodesic_features(x, ei, n, d.anchor_nodes)
        if not np.array_equal(out.numpy().view(np.uint32), want.view(np.uint32)):
            bad += 1
            print("MISMATCH at call", it, mode, flush=True)
    # ordinary pageable traffic on recycled host memory, like the model's .to(device) that died in round 3
    junk = torch.rand(1 << 22)
    dj = junk.to(dev)
    back = dj.cpu()
    assert torch.equal(back, junk)
    del out, junk, dj, back, d
    if it % 7 == 0:
        gc.collect()
    if it % 40 == 0:
        torch.cuda.synchronize()
        print(it, mode, "rss MB %.0f" % (proc.memory_info().rss / 1e6), "cuda MB %.0f" % (torch.cuda.memory_allocated() / 1e6), "fds", proc.num_fds(), flush=True)
os.environ.pop("GRAPHPOPE_HOST_RESULT", None)
gp.clear_cache()
torch.cuda.synchronize()
print("done: 240 calls in %.1f s, %d mismatches, rss MB %.0f, fds %d" % (time.time() - t0, bad, proc.memory_info().rss / 1e6, proc.num_fds()))
sys.exit(1 if bad else 0)
