#!/usr/bin/env python3
"""How fast are the library GEMMs (torch -> hipBLASLt / rocBLAS, fp32) on the SAGE layer-0 shapes?  (GPU box)"""
import torch, time
dev = torch.device("cuda")
torch.backends.cuda.matmul.allow_tf32 = False
def timed(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
n_dst, c_in, c_out = 9988, 756, 256
agg = torch.randn(n_dst, c_in, device=dev); xd = torch.randn(n_dst, c_in, device=dev)
wl = torch.randn(c_out, c_in, device=dev); wr = torch.randn(c_out, c_in, device=dev); b = torch.randn(c_out, device=dev)
g = torch.randn(n_dst, c_out, device=dev)
cat = torch.cat([agg, xd], 1); wcat = torch.cat([wl, wr], 1)
t = timed(lambda: torch.addmm(b, cat, wcat.t()))
print(f"fwd  [9988x1512]x[1512x256] addmm: {t:.1f} us  {2*n_dst*2*c_in*c_out/t/1e6:.1f} TF")
t = timed(lambda: (agg @ wl.t()).add_(xd @ wr.t()))
print(f"fwd  two mm + add: {t:.1f} us")
t = timed(lambda: g.t() @ agg)
print(f"wgrad [256x9988]x[9988x756]: {t:.1f} us  {2*n_dst*c_in*c_out/t/1e6:.1f} TF")
t = timed(lambda: g.t() @ cat)
print(f"wgrad both in one [256x9988]x[9988x1512]: {t:.1f} us  {2*n_dst*2*c_in*c_out/t/1e6:.1f} TF")
