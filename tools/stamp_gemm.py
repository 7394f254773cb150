#!/usr/bin/env python3
"""Diagnostic (GPU box): stage timeline of the SAGE forward GEMM (layer 0 shape) with the -DPOPE_STAMP build."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libgraphpope_hip_stamp.so")
from graphpope_amd import engine
from graphpope_amd.sage import SAGEConv, SampledAdj
lib = _lib.load(); dev = engine.require_gpu()
n_dst, n_src, c_in, c_out = 9988, 37799, 756, 256
rs = np.random.RandomState(0)
deg = rs.randint(1, 11, size=n_dst); rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
col = rs.randint(0, n_src, size=int(rowptr[-1])).astype(np.int32)
adj = SampledAdj(torch.tensor(rowptr), torch.tensor(col), n_src).to(dev)
conv = SAGEConv(c_in, c_out).to(dev); x = torch.randn(n_src, c_in, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if mode == "fwd":
    with torch.no_grad():
        for _ in range(5): conv((x, x[:n_dst]), adj)
else:                                   # last GEMM of the backward = grad_w_r (reduction over the n_dst rows, split-K)
    for _ in range(3):
        out = conv((x, x[:n_dst]), adj); out.sum().backward()
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
lib.pope_debug_read_gemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.pope_debug_read_gemm_stamps(buf.ctypes.data, buf.size)
st = buf.reshape(8192, 8).astype(np.int64); st = st[st[:, 0] > 0]
d = np.diff(st[:, :6], axis=1) / 100.0
names = ["wait barrier 1", "LDS stores", "wait barrier 2", "issue next loads", "fragment reads + MFMA"]
print(f"{mode}: {len(st)} waves; stage 2; microseconds (median / p90):")
for i, nm in enumerate(names): print(f"  {nm:24s} {np.median(d[:, i]):6.2f} / {np.percentile(d[:, i], 90):6.2f}")
print(f"  stage total              {np.median(d.sum(1)):6.2f}")
t0 = st[:, 0].min(); print("  stage-2 start spread (us): p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile((st[:, 0] - t0) / 100.0, [10, 50, 90, 100])))
