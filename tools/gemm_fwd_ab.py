#!/usr/bin/env python3
"""A/B of the SAGE layer-0 forward projection on the config-2 block (GPU box): stream-K against the whole-tile kernels and
torch.addmm (hipBLASLt).  python tools/gemm_fwd_ab.py"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib, engine
lib = _lib.load(); dev = engine.require_gpu()
torch.manual_seed(0)
shapes = [(9988, 756, 256), (10300, 756, 256), (9000, 756, 256), (40300, 756, 256), (9988, 532, 256), (19717, 532, 256)] if len(sys.argv) < 2 else [tuple(int(v) for v in sys.argv[1:4])]
for n_dst, c_in, c_out in shapes:
    agg = torch.rand(n_dst, c_in, device=dev); xd = torch.rand(n_dst, c_in, device=dev)
    wl = torch.randn(c_out, c_in, device=dev) * 0.05; wr = torch.randn(c_out, c_in, device=dev) * 0.05; b = torch.randn(c_out, device=dev)
    rowptr = torch.zeros(n_dst + 1, dtype=torch.int32, device=dev); col = torch.zeros(4, dtype=torch.int32, device=dev)     # empty block: agg = 0
    out = torch.empty(n_dst, c_out, device=dev); aggb = torch.empty_like(agg)
    nbytes = lib.sage_conv_forward_scratch_bytes(n_dst, c_in, c_out)
    scratch = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    want = xd.double() @ wr.double().t() + b.double()          # agg = 0 for the empty block
    res = {}
    for name, tile, small in (("auto", 0, 0), ("stream-K", 7, 0), ("64x64", 1, 0), ("small-tile16", 0, 1), ("auto", 0, 0), ("small-tile16", 0, 1)):
        lib.pope_debug_set(_lib.KNOB_GEMM_TILE, tile)
        lib.pope_debug_set(_lib.KNOB_GEMM_SMALL_TILE16, small)
        lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 0)
        def run():
            _lib.check(lib.sage_conv_forward(_lib.ptr(rowptr), _lib.ptr(col), n_dst, n_dst, 0, _lib.ptr(xd), c_in, _lib.ptr(wl), _lib.ptr(b), _lib.ptr(wr),
                                             c_out, _lib.ptr(aggb), _lib.ptr(out), _lib.ptr(scratch), nbytes, None, stream))
        for _ in range(3): run()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20): run()
        ev[1].record(); torch.cuda.synchronize()
        err = float((out.double() - want).abs().max())
        res.setdefault(name, []).append((ev[0].elapsed_time(ev[1]) / 20 * 1e3, err))
    lib.pope_debug_set(_lib.KNOB_GEMM_TILE, 0)
    lib.pope_debug_set(_lib.KNOB_GEMM_SMALL_TILE16, 1)
    lib.pope_debug_set(_lib.KNOB_SAGE_FORWARD_OVERLAP, 1)
    xcat = torch.cat([aggb, xd], 1); wcat = torch.cat([wl, wr], 1)
    for _ in range(3): torch.addmm(b, xcat, wcat.t())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): torch.addmm(b, xcat, wcat.t())
    ev[1].record(); torch.cuda.synchronize()
    lib_us = ev[0].elapsed_time(ev[1]) / 20 * 1e3
    flops = 4.0 * n_dst * c_in * c_out
    print(f"M={n_dst} K=2x{c_in} N={c_out}  (gather of an empty block included: ~{n_dst*c_in*4/4e6:.0f} us-ish fill)  hipBLASLt {lib_us:.1f} us")
    for k, runs in res.items():
        for us, err in runs:
            print(f"   {k:13s} {us:8.1f} us  {flops/us/1e6:6.1f} TF (whole call)  max err {err:.2e}")
