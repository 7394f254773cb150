import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphpope_amd import engine, synth
dev = engine.require_gpu()
n, d, k = synth.FLICKR_N, 128, 256
emb = torch.randn(n, d, device=dev); x = torch.rand(n, 500, device=dev)
anchors = synth.seeded_anchors(n, k, 42)
for _ in range(10): engine.pairwise_features(x, emb, anchors, "euclidean")
torch.cuda.synchronize()
