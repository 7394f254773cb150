#!/usr/bin/env python3
"""Is the SAGE training step GPU-bound or launch-bound?  (GPU box)  Wall per step, host enqueue time per step, and the
same with torch's own BN/ReLU/dropout for comparison.  Prints one JSON object."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth, sage
from graphpope_amd.optim import Adam
import torch.nn.functional as F

dev = engine.require_gpu()
ei_np, n = synth.flickr_like()
feats = torch.rand(n, 756, device=dev)
rowptr = np.concatenate([[0], np.cumsum(np.bincount(ei_np[0], minlength=n))])
rng = np.random.default_rng(0)
batches = []
for b in range(8):
    seeds = rng.choice(n, 1550, replace=False)
    n_id, adjs = sage.sample_batch(rowptr, ei_np[1], seeds, sizes=(25, 10), rng=rng)
    batches.append((torch.as_tensor(n_id, device=dev), [a.to(dev) for a in adjs], torch.randint(0, 7, (1550,), device=dev)))


FUSED_ADAM = False


def run(fused, steps=100, adam=False):
    global FUSED_ADAM
    FUSED_ADAM = adam
    torch.manual_seed(0)
    model = sage.SAGE(756, 7, 256, 3).to(dev)
    if not fused:
        def fwd(x, adjs):
            for i, adj_t in enumerate(adjs):
                x = model.convs[i]((x, x[:adj_t.size(0)]), adj_t)
                if i < len(adjs) - 1:
                    x = F.dropout(model.bns[i](x).relu_(), p=0.5, training=True)
            return x
    else:
        fwd = model
    opt = Adam(model.parameters(), lr=1e-3) if FUSED_ADAM else torch.optim.Adam(model.parameters(), lr=1e-3)

    def step(i):
        n_id, adjs, y = batches[i % 8]
        x = feats.index_select(0, n_id)
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(fwd(x, adjs), y)
        loss.backward()
        opt.step()
    for i in range(10): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps): step(i)
    t_host = (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / steps
    return {"wall_ms": t_wall * 1e3, "host_enqueue_ms": t_host * 1e3}

if len(sys.argv) > 1:
    res = {sys.argv[1]: run(sys.argv[1] == "fused", adam=True)}
else:
    res = {"fused": run(True), "torch_bn": run(False), "fused+fused_adam": run(True, adam=True), "torch_bn+fused_adam": run(False, adam=True)}
print(json.dumps(res))
