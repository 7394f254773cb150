"""On-disk cache of hop planes.

File ``<dir>/pope_<key>.npz``: ``planes`` int64 [1 + bits, N, W] (the library's bit-sliced format, include/graphpope_hip.h),
``n_hop_bits``, ``anchors`` int64 [K], ``num_nodes``.  ``key`` = SHA-1 over (N, E, anchors, a 128-bit digest of
edge_index computed on the device), so a different graph, edge order or anchor draw never hits.
The planes are (1 + bits) bits per embedding: Flickr / 256 anchors is 11.4 MB against 91 MB of float32.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np
import torch


def graph_key(edge_index: torch.Tensor, num_nodes: int, anchors: np.ndarray) -> str:
    e = edge_index.reshape(-1).to(torch.int64)
    idx = torch.arange(e.numel(), device=e.device, dtype=torch.int64)
    # four position-dependent 64-bit sums (wrap-around arithmetic): cheap on the device, sensitive to order and content
    mixed = (e + (0x9E3779B97F4A7C15 - (1 << 64))) * (2 * idx + 1)
    digest = [int(mixed.sum().item()), int((mixed ^ (mixed >> 29)).sum().item()),
              int((e * (idx % 1000003 + 7)).sum().item()), int(e.sum().item())]
    h = hashlib.sha1()
    h.update(np.asarray([num_nodes, e.numel()] + digest, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(anchors, dtype=np.int64).tobytes())
    return h.hexdigest()[:24]


def _path(cache_dir: str, key: str) -> str:
    return os.path.join(cache_dir, f"pope_{key}.npz")


def save(cache_dir: str, key: str, hp, anchors) -> str:
    os.makedirs(cache_dir, exist_ok=True)
    path, tmp = _path(cache_dir, key), _path(cache_dir, key) + f".{os.getpid()}.tmp.npz"
    np.savez(tmp, planes=hp.valid().cpu().numpy(), n_hop_bits=np.int64(hp.n_hop_bits),
             anchors=np.asarray(anchors, dtype=np.int64), num_nodes=np.int64(hp.num_nodes))
    os.replace(tmp, path)                               # atomic: concurrent ranks never read a half-written file
    return path


def load(cache_dir: str, key: str, device):
    path = _path(cache_dir, key)
    if not os.path.exists(path):
        return None
    with np.load(path) as z:
        planes = torch.from_numpy(z["planes"]).to(device)
        return planes, int(z["n_hop_bits"])
