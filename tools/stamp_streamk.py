#!/usr/bin/env python3
"""Where does a stage of k_gemm_streamk go?  (GPU box)   make -C graphpope_amd/csrc stamp && python tools/stamp_streamk.py
Stamps (shader clock) of stage s_begin + 6 of every block's first segment, waves 0 and 4:
  0 top of stage | 1 after the vmcnt wait | 2 after the barrier | 3 after the first two MFMAs were issued | 4 after the last MFMA was issued"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libgraphpope_hip_stamp.so")
from graphpope_amd import engine
lib = _lib.load(); dev = engine.require_gpu()
lib.pope_debug_set(_lib.KNOB_GEMM_TILE, int(os.environ.get('SK_TILE', '4')))
n_dst, c_in, c_out = 9988, 756, 256
torch.manual_seed(0)
xd = torch.rand(n_dst, c_in, device=dev); wl = torch.randn(c_out, c_in, device=dev) * 0.05; wr = torch.randn(c_out, c_in, device=dev) * 0.05
b = torch.randn(c_out, device=dev); rowptr = torch.zeros(n_dst + 1, dtype=torch.int32, device=dev); col = torch.zeros(4, dtype=torch.int32, device=dev)
out = torch.empty(n_dst, c_out, device=dev); agg = torch.empty(n_dst, c_in, device=dev)
nbytes = lib.sage_conv_forward_scratch_bytes(n_dst, c_in, c_out); scratch = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    _lib.check(lib.sage_conv_forward(_lib.ptr(rowptr), _lib.ptr(col), n_dst, n_dst, 0, _lib.ptr(xd), c_in, _lib.ptr(wl), _lib.ptr(b), _lib.ptr(wr), c_out,
                                     _lib.ptr(agg), _lib.ptr(out), _lib.ptr(scratch), nbytes, stream))
torch.cuda.synchronize()
buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
lib.pope_debug_read_streamk_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pope_debug_read_streamk_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(256, 8, 8).astype(np.int64)
print("tile knob", os.environ.get("SK_TILE", "4"))
print("slots: knob 4-7: 0 top | 1 after vmcnt wait | 2 after barrier | 3 first group issued | 4 last MFMA issued")
print("       knob 10 : consumer 0 top | 1 first group issued | 2 last MFMA issued | 3 after barrier;  loader 0 top | 1 DMAs issued | 2 landed | 3 after barrier")
t0 = st[:, :, 0].min(axis=1, keepdims=True)                      # per block: earliest top-of-stage stamp
for w in range(8):
    if not st[:, w, 0].any():
        continue
    rel = st[:, w, :5] - t0
    print(f"wave {w}: median stamps relative to the block's earliest wave: " + "  ".join(f"{np.median(rel[:, i]):7.0f}" for i in range(5)))
