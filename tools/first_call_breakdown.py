#!/usr/bin/env python3
"""Where the FIRST host -> host Graphpope call of a process spends its time (GPU box): python tools/first_call_breakdown.py

The reference memoises (utils.py:195-208), so a process makes this call once: the cold figure is the real one.
Every line is one cold measurement in a fresh process, in the order the call meets them.
"""
import ctypes
import os
import sys
import time

T0 = time.perf_counter()
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ms(t0):
    return (time.perf_counter() - t0) * 1e3


print(f"import numpy + torch:                                   {ms(T0):9.2f} ms")
for f in ("enabled", "defrag", "shmem_enabled"):
    try:
        print(f"THP {f}: " + open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip())
    except OSError as exc:
        print("THP", f, exc)
print("cpus:", len(os.sched_getaffinity(0)))

t0 = time.perf_counter()
from graphpope_amd import _lib  # noqa: E402
if os.environ.get("GRAPHPOPE_DIAG_LIB"):                     # A/B of two library builds: one per process
    _lib.LIB_PATH = os.environ["GRAPHPOPE_DIAG_LIB"]
from graphpope_amd import engine, synth  # noqa: E402
from graphpope_amd import utils as gp  # noqa: E402
lib = _lib.load()
print(f"import graphpope_amd + dlopen:                          {ms(t0):9.2f} ms")
t0 = time.perf_counter()
dev = engine.require_gpu()
torch.cuda.current_stream().synchronize()
print(f"require_gpu (HIP runtime init):                         {ms(t0):9.2f} ms")

ei_np, n = synth.flickr_like()
F, K = 500, 256
x = torch.rand(n, F)
ei_cpu = torch.as_tensor(ei_np)
anchors = synth.seeded_anchors(n, K, 42)
nbytes = n * (F + K) * 4


def sync():
    torch.cuda.current_stream().synchronize()


mode = sys.argv[1] if len(sys.argv) > 1 else "pieces"
if mode == "pieces":
    t0 = time.perf_counter(); a = torch.empty(1 << 20, dtype=torch.uint8, device=dev); sync()
    print(f"first device allocation (1 MB):                         {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); p = torch.empty(16 << 20, dtype=torch.uint8, pin_memory=True)
    print(f"pinned 16 MB, cold:                                     {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); p2 = torch.empty(32 << 20, dtype=torch.uint8, pin_memory=True)
    print(f"pinned 32 MB, cold:                                     {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); out_pin = torch.empty((n, F + K), dtype=torch.float32, pin_memory=True)
    print(f"pinned result {nbytes / 1e6:.0f} MB, cold:                            {ms(t0):9.2f} ms")
    del out_pin
    t0 = time.perf_counter(); out_pin = torch.empty((n, F + K), dtype=torch.float32, pin_memory=True)
    print(f"pinned result again (torch's host cache):               {ms(t0):9.2f} ms")
    for th in (1, 8, 16):
        t0 = time.perf_counter(); out = torch.empty((n, F + K), dtype=torch.float32)
        a0 = ms(t0)
        t0 = time.perf_counter(); engine.host_copy_2d(x, out[:, :F], threads=th)
        print(f"pageable result: alloc {a0:6.2f} ms, first-touch x copy, {th:2d} threads: {ms(t0):9.2f} ms")
        t0 = time.perf_counter(); engine.host_copy_2d(x, out[:, :F], threads=th)
        print(f"   the same copy again (pages present):                 {ms(t0):9.2f} ms")
        del out
    # madvise(MADV_HUGEPAGE) before the first touch
    libc = ctypes.CDLL("libc.so.6", use_errno=True)
    out = torch.empty((n, F + K), dtype=torch.float32)
    lo = (out.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)
    hi = (out.data_ptr() + nbytes) & ~((1 << 21) - 1)
    t0 = time.perf_counter(); rc = libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), 14)
    print(f"madvise(MADV_HUGEPAGE) rc={rc} errno={ctypes.get_errno()}:                     {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); engine.host_copy_2d(x, out[:, :F], threads=16)
    print(f"   first-touch x copy after it, 16 threads:             {ms(t0):9.2f} ms")
    # registering the pageable result for DMA
    t0 = time.perf_counter(); rc = torch.cuda.cudart().cudaHostRegister(out.data_ptr(), nbytes, 0)
    print(f"hipHostRegister of the touched pageable result rc={int(rc)}:  {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(out.data_ptr())
    print(f"hipHostUnregister:                                      {ms(t0):9.2f} ms")
    del out
    t0 = time.perf_counter(); ei = engine.stage_to_device(ei_cpu, dev); sync()
    print(f"edge_index up, cold (pinned staging 14 MB + H2D):       {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); ei = engine.stage_to_device(ei_cpu, dev); sync()
    print(f"edge_index up, warm:                                    {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); emb = engine.geodesic_features(None, ei, n, anchors, shard=False); sync()
    print(f"first geodesic_features (code load, workspace, slots):  {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); emb = engine.geodesic_features(None, ei, n, anchors, shard=False); sync()
    print(f"second geodesic_features:                               {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); engine.copy_columns_to_host(emb, out_pin[:, F:]); sync()
    print(f"pitched D2H into the pinned result, first:              {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); engine.copy_columns_to_host(emb, out_pin[:, F:]); sync()
    print(f"pitched D2H into the pinned result, again:              {ms(t0):9.2f} ms")
    out = torch.empty((n, F + K), dtype=torch.float32)
    engine.host_copy_2d(x, out[:, :F], threads=16)
    t0 = time.perf_counter(); engine.copy_columns_to_host(emb, out[:, F:]); sync()
    print(f"pitched D2H into a PAGEABLE (touched) result:           {ms(t0):9.2f} ms")
    t0 = time.perf_counter(); engine.copy_columns_to_host(emb, out[:, F:]); sync()
    print(f"   again:                                               {ms(t0):9.2f} ms")
    lin = torch.empty((n, K), dtype=torch.float32)
    t0 = time.perf_counter(); lin.copy_(emb); sync()
    print(f"contiguous D2H into pageable [N, K] (torch copy_):      {ms(t0):9.2f} ms")
else:
    class Data:
        pass
    d = Data()
    d.x, d.edge_index, d.num_nodes = x, ei_cpu, n
    import contextlib
    for i in range(4):
        gp.clear_cache()
        np.random.seed(42)
        with contextlib.redirect_stdout(sys.stderr):
            t0 = time.perf_counter(); out = gp.Graphpope(d, "flickr", "geodesic", "stochastic", K, None, 6); dt = ms(t0)
        print(f"Graphpope call {i}: {dt:9.2f} ms  pinned={out.is_pinned()}")
        del out
