#!/usr/bin/env python3
"""Which part of the training step survives HIP graph capture (GPU box): python tools/capture_bisect.py
Every case runs in a subprocess of its own, so a crash in one does not hide the others."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["sampler", "forward", "fwd_bwd_nolanes", "fwd_bwd_lanes", "full_nolanes", "full", "full_sampler"]

if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=300)
        tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
        print(f"{c:18s} rc={r.returncode}  " + " | ".join(tail), flush=True)
    sys.exit(0)

case = sys.argv[1]
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from graphpope_amd import _lib, engine, synth  # noqa: E402
from graphpope_amd.optim import Adam  # noqa: E402
from graphpope_amd.sage import SAGE, IndexedFeatures, cross_entropy  # noqa: E402
from graphpope_amd.sampler import NeighborSampler  # noqa: E402
from graphpope_amd.train import SageTrainStep, StepState  # noqa: E402

dev = engine.require_gpu()
lib = _lib.load()
ei = synth.powerlaw_graph(6000, 40000, seed=7, alpha=0.9, shift=0.8)
csr = engine.build_csr(torch.as_tensor(ei, device=dev), 6000)
feats = torch.randn(6000, 40, device=dev)
labels = torch.randint(0, 5, (6000,), device=dev)
sampler = NeighborSampler(csr.rowptr, csr.col, 6000, (25, 10))
seeds = torch.arange(256, device=dev)
y = labels[:256].contiguous()
torch.manual_seed(0)
model = SAGE(40, 5, 48, 3).to(dev)
opt = Adam(model.parameters(), lr=0.01)
state = StepState(dev, 1)
batch = sampler.sample_device(seeds, seed=3)
one = torch.ones((), device=dev)
if "nolanes" in case:
    lib.pope_debug_set(_lib.KNOB_SAGE_LANES, 0)


def body():
    if case == "sampler":
        sampler.sample_device(seeds, seed=0, out=batch, seed_dev=state.sample_seed)
        return
    model.dropout_seed_dev = state.dropout_seed
    out = model(IndexedFeatures(feats, batch.n_id), batch.adjs)
    if case == "forward":
        return
    loss = cross_entropy(out, y)
    for p in model.parameters():
        p.grad = None
    loss.backward(gradient=one)
    if case.startswith("fwd_bwd"):
        return
    opt.use_device_step(state.adam_step)
    opt.step()
    state.advance()


if case == "full_sampler":
    st = SageTrainStep(model, opt, feats, 256, sampler=sampler, clip=0.5)
    for i in range(5):
        print("loss", st.step(seeds, y).item())
    sys.exit(0)

with torch.autograd.set_multithreading_enabled(False):
    for _ in range(2):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
print("captured and replayed:", case)
