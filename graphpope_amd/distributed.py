"""Anchor sharding over the GPUs of one node + all-gather of the hop planes.

The reference has no collective on this path: every Lightning DDP rank recomputes all K anchors
(/root/reference/main.py:94-98 runs per process).  Here rank g runs the multi-source BFS for the
contiguous anchor slice g and the bit-sliced hop planes -- (1 + hop bits) bits per (node, anchor)
instead of a 32-bit float -- are exchanged with ONE ``all_gather`` (RCCL over xGMI when the backend
is "nccl"); every rank then expands them into the same [N, F+K] matrix.

This module is device-agnostic host logic (the BFS and the expansion are passed in), so the N > 1
path is covered by world_size-2 ``gloo`` tests on CPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(group)


def rank(group=None) -> int:
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    return dist.get_rank(group)


def shard_size(k: int, world: int) -> int:
    """Anchors per rank: ceil(K / world); shard g owns anchors [g*size, min((g+1)*size, K)) in draw order."""
    return -(-k // world)


def shard_anchors(anchors: np.ndarray, world: int, rnk: int):
    """(padded anchors of length shard_size, number of real anchors) for one rank.

    All shards must have the same shape for the all-gather, so a short (or empty) last shard is
    padded by repeating an anchor; the padded columns fall beyond column K and are dropped.
    """
    anchors = np.asarray(anchors, dtype=np.int64)
    size = shard_size(anchors.size, world)
    lo = min(rnk * size, anchors.size)
    hi = min(lo + size, anchors.size)
    real = anchors[lo:hi]
    pad_value = real[-1] if real.size else anchors[0]
    padded = np.concatenate([real, np.full(size - real.size, pad_value, dtype=np.int64)])
    return padded, int(real.size)


def _all_gather_begin(local: torch.Tensor, group):
    """Start the all-gather of one [..] tensor per rank; returns (gathered [world, ..], work).

    With RCCL the collective runs on the communicator's stream once the producer of `local` on the current stream is
    done: whatever the caller enqueues next on the current stream runs UNDERNEATH the exchange; ``work.wait()`` makes
    the current stream (not the host) wait for the gathered planes.
    """
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "nccl":
        work = dist.all_gather_into_tensor(out, local, group=group, async_op=True)   # one RCCL all-gather, no staging copies
    else:
        work = dist.all_gather(list(out.unbind(0)), local, group=group, async_op=True)
    return out, work


def _exchange_and_expand(local, group, x, f, num_nodes, cols, bits, size, finalize_fn, finalize_all_fn, copy_x_fn):
    """all-gather the planes; copy the features into ``out`` while they travel; expand every shard's columns."""
    gathered, work = _all_gather_begin(local, group)                 # [world, 1 + bits, N, W]
    out = torch.empty((num_nodes, cols), dtype=torch.float32, device=x.device)
    if copy_x_fn is not None:
        copy_x_fn(x, f, out)                 # out[:, :F] = x: 2/3 of the expansion's HBM traffic, hidden under the xGMI exchange
        x = None
    work.wait()
    if finalize_all_fn is not None:
        finalize_all_fn(gathered, bits, num_nodes, size, x, f, out)
    else:
        for g in range(gathered.shape[0]):
            finalize_fn(gathered[g], bits, num_nodes, size, x if g == 0 else None, f, out, g * size)
    return out


SPECULATIVE_HOP_BITS = 4


SPECULATIVE_LEVELS = 12           # levels begin_fn enqueues without waiting (engine.SPECULATIVE_LEVELS)


def _speculative(x, num_nodes, anchors, group, begin_fn, finalize_all_fn, copy_x_fn):
    """Fast path with no host synchronisation in the middle: enqueue the BFS, all-gather seen + 4 hop-bit planes and
    expand them, THEN look at the verdicts.  Returns the matrix, or None if some rank needs more than 4 hop bits (or its
    edge list was not sorted): the caller then takes the general path.

    If the pending BFS offers ``verdict()`` (a tiny device tensor), the verdicts travel with the planes -- a second
    all-gather of 8 bytes per rank on the same communicator -- and every rank reads ALL of them with its one
    synchronisation at the end: no all-reduce, no second host wait."""
    world = dist.get_world_size(group)
    k, f = int(anchors.size), x.shape[1]
    size = shard_size(k, world)
    local_anchors, _ = shard_anchors(anchors, world, dist.get_rank(group))
    pending = begin_fn(local_anchors)
    verdicts = None
    if hasattr(pending, "verdict"):
        verdicts = _all_gather_begin(pending.verdict(), group)         # ([world, 2] int32, work)
    cols = f + world * size
    out = _exchange_and_expand(pending.speculative_planes(), group, x, f, num_nodes, cols, SPECULATIVE_HOP_BITS, size,
                               None, finalize_all_fn, copy_x_fn)       # [world, 5, N, W] slices: no staging copy
    if verdicts is not None:
        verdicts[1].wait()
        v = verdicts[0].cpu()                                         # the first host synchronisation of the call
        failed = bool(((v[:, 1] != 0) | (v[:, 0] >= SPECULATIVE_LEVELS)).any())
    else:
        hp = pending.finish()                                         # the first host synchronisation of the call
        ok = torch.tensor([0 if (hp is not None and hp.n_hop_bits <= SPECULATIVE_HOP_BITS) else 1], dtype=torch.int32, device=x.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MAX, group=group)
        failed = bool(int(ok.item()))
    if failed:
        return None
    if cols != f + k:
        out = out[:, : f + k].contiguous()
    return out


def sharded_geodesic_features(x: torch.Tensor, num_nodes: int, anchors: np.ndarray, group, bfs_fn, finalize_fn,
                              finalize_all_fn=None, begin_fn=None, copy_x_fn=None) -> torch.Tensor:
    """Every rank returns the full [N, F+K] float32 matrix.

    bfs_fn(anchors) -> object with .planes ([>= 1 + n_hop_bits, N, W] int64), .n_hop_bits
    finalize_fn(planes, n_hop_bits, N, K_shard, x_or_None, F, out, c0) writes x and one shard's columns;
    finalize_all_fn(gathered, n_hop_bits, N, K_shard, x, F, out), if given, writes every shard in one pass instead.
    begin_fn(anchors) -> object with .speculative_planes() and .finish(): enables the synchronisation-free fast path.
    copy_x_fn(x, F, out), if given, writes out[:, :F] = x on its own; it is enqueued right after the all-gather has
    been started, so the feature copy overlaps the exchange and the expansion afterwards only writes the K columns.
    """
    world, rnk = dist.get_world_size(group), dist.get_rank(group)
    anchors = np.asarray(anchors, dtype=np.int64)
    if begin_fn is not None and finalize_all_fn is not None:
        out = _speculative(x, num_nodes, anchors, group, begin_fn, finalize_all_fn, copy_x_fn)
        if out is not None:
            return out
    k, f = int(anchors.size), x.shape[1]
    size = shard_size(k, world)
    local_anchors, _ = shard_anchors(anchors, world, rnk)
    hp = bfs_fn(local_anchors)

    # Ranks may have found different depths: agree on the number of hop-bit planes to exchange.
    bits_t = torch.tensor([hp.n_hop_bits], dtype=torch.int32, device=x.device)
    dist.all_reduce(bits_t, op=dist.ReduceOp.MAX, group=group)
    bits = int(bits_t.item())
    local = torch.zeros((1 + bits,) + tuple(hp.planes.shape[1:]), dtype=hp.planes.dtype, device=hp.planes.device)
    local[: 1 + hp.n_hop_bits] = hp.planes[: 1 + hp.n_hop_bits]
    cols = f + world * size
    out = _exchange_and_expand(local, group, x, f, num_nodes, cols, bits, size, finalize_fn, finalize_all_fn, copy_x_fn)
    if cols != f + k:                                               # K not divisible by world: drop the padding
        out = out[:, : f + k].contiguous()
    return out
