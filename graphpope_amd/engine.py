"""Device-side engine: torch owns memory and streams, libgraphpope_hip.so does the work.

Every function here takes/returns tensors that live on the MI355X; ``graphpope_amd.utils`` wraps
them behind the reference's ``Graphpope`` call.  PyTorch is plumbing only (allocation, streams,
``torch.distributed``): no torch op computes any part of the embedding.
"""
from __future__ import annotations

import ctypes
import os
import sys
import threading
import time
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

DEFAULT_PLANE_CAPACITY = 8          # hop counts < 256; grown on POPE_ERR_HOP_OVERFLOW


def require_gpu(device=None) -> torch.device:
    """The product path needs a GPU and the HIP library; it never computes on the CPU."""
    lib = _lib.load()
    if not torch.cuda.is_available() or lib.pope_require_device(None) != _lib.OK:
        raise RuntimeError("graphpope_amd needs an MI355X (no gfx950 device visible: %s); there is no CPU fallback"
                           % (lib.pope_last_error().decode() or "torch.cuda.is_available() is False"))
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(f"graphpope_amd computes on the GPU only, got device {dev}")
    return dev


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bytes(n: int, device) -> torch.Tensor:
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------
# host <-> device plumbing of the host -> host Graphpope call (utils.py:129-147): only edge_index goes up and only the K
# embedding columns come down; data.x is copied host to host into the result by pope_host_copy_2d.
# ------------------------------------------------------------------------------------------------
def host_threads() -> int:
    """Host threads for the feature copy: GRAPHPOPE_HOST_THREADS, else the cores this process may run on (at most 16)."""
    import os
    env = os.environ.get("GRAPHPOPE_HOST_THREADS")
    if env:
        return max(1, int(env))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, 16))


def host_copy_2d(src: torch.Tensor, dst: torch.Tensor, threads: int = 0) -> None:
    """dst[:, :] = src for two HOST matrices whose rows are contiguous (row pitches may differ); blocking, multi-threaded."""
    lib = _lib.load()
    assert not src.is_cuda and not dst.is_cuda and src.dim() == 2 and src.shape == dst.shape and src.dtype == dst.dtype
    if src.numel() == 0:
        return
    assert src.stride(1) == 1 and dst.stride(1) == 1
    es = src.element_size()
    check(lib.pope_host_copy_2d(ptr(src), src.stride(0) * es, ptr(dst), dst.stride(0) * es, src.shape[1] * es, src.shape[0],
                                threads or host_threads()))


def stage_to_device(t: torch.Tensor, device, register: bool = True) -> torch.Tensor:
    """A pageable host tensor -> device, asynchronously on the current stream.

    The caller's pages are registered with the HIP runtime for the length of the copy (pope_host_pin: microseconds per
    MB, no staging copy, no pinned allocation that outlives the call) and the DMA reads them in place; the registration
    is dropped once the copy has completed.  If the runtime refuses the registration the tensor goes through pinned
    staging memory instead (threaded host copy + one DMA)."""
    if t.is_cuda:
        return t.to(device)
    t = t.contiguous()
    if t.is_pinned() or t.numel() == 0:
        return t.to(device, non_blocking=True)
    lib = _lib.load()
    nbytes = t.numel() * t.element_size()
    if register and nbytes >= (1 << 20) and lib.pope_host_pin(ptr(t), nbytes) == _lib.OK:
        try:
            with torch.cuda.device(device):
                out = torch.empty(t.shape, dtype=t.dtype, device=device)
                check(lib.pope_copy_to_device(ptr(t), ptr(out), nbytes, _stream()))
                torch.cuda.current_stream().synchronize()          # the pages stay registered exactly as long as the DMA reads them
        except BaseException:
            lib.pope_host_unpin(ptr(t))                             # the copy's own error is the one to report
            raise
        check(lib.pope_host_unpin(ptr(t)))                          # a refused release is an error: the pages would stay registered
        return out
    staged = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host_copy_2d(t.view(1, -1), staged.view(1, -1))
    return staged.to(device, non_blocking=True)


class staged:
    """``with staged(t, device) as t_dev:`` -- stage_to_device without the wait: the copy is enqueued, the body runs
    (enqueueing the GPU work that consumes t_dev), and the caller's pages are released on exit, once the event recorded
    behind the copy has completed (normally long before: the body has waited for results that depend on it)."""

    def __init__(self, t: torch.Tensor, device, register: bool = True):
        self.t, self.device, self.pinned, self.event, self.register = t, device, False, None, register

    def __enter__(self) -> torch.Tensor:
        t = self.t
        if t.is_cuda or t.numel() == 0 or t.is_pinned():
            return stage_to_device(t, self.device)
        lib = _lib.load()
        self.t = t = t.contiguous()
        nbytes = t.numel() * t.element_size()
        if not self.register or nbytes < (1 << 20) or lib.pope_host_pin(ptr(t), nbytes) != _lib.OK:
            return stage_to_device(t, self.device, register=False)
        self.pinned = True
        try:
            with torch.cuda.device(self.device):
                out = torch.empty(t.shape, dtype=t.dtype, device=self.device)
                check(lib.pope_copy_to_device(ptr(t), ptr(out), nbytes, _stream()))
                self.event = torch.cuda.Event()
                self.event.record()
        except BaseException:
            self.__exit__(None, None, None)
            raise
        return out

    def __exit__(self, *exc):
        if self.pinned:
            if self.event is not None:
                self.event.synchronize()
            self.pinned = False
            rc = _lib.load().pope_host_unpin(ptr(self.t))
            if rc != _lib.OK and exc[0] is None:                   # a refused release leaves the caller's pages registered: do not hide it
                check(rc)
        return False


# The returned [N, F+K] tensor lives in ordinary pageable memory.  Its pages come from an anonymous private mapping that is
# handed back here when the tensor dies and reused by the next call of the same size: a repeated call then takes no page
# faults (66 000 of them, or 135 huge ones whose compaction was seen to stall the GPU queues of a large process for
# 5-15 ms afterwards), and a process that calls once -- the reference memoises -- pays them once, as with torch.empty.
_RESULT_POOL: dict = {}              # nbytes -> mmap object of a result that has been freed (at most one entry)
_RESULT_POOL_LOCK = threading.Lock()


def _result_pool_cap() -> int:
    return int(float(os.environ.get("GRAPHPOPE_RESULT_POOL_MB", "1024")) * (1 << 20))


def _give_back(mm, nbytes: int) -> None:
    with _RESULT_POOL_LOCK:
        if nbytes <= _result_pool_cap() and not _RESULT_POOL:
            _RESULT_POOL[nbytes] = mm
    # otherwise the last reference is dropped here and the mapping is unmapped with it


def host_result_tensor(rows: int, cols: int, with_origin: bool = False):
    """An uninitialised contiguous float32 [rows, cols] HOST tensor in pageable memory (not pinned, not shared), backed by
    the one-entry pool above.  GRAPHPOPE_RESULT_POOL_MB caps what the pool may keep (default 1024; 0: plain torch.empty).
    ``with_origin``: (tensor, reused) -- reused = the pages come from the pool, i.e. writing them takes no page faults."""
    t, reused = _host_result_tensor(rows, cols)
    return (t, reused) if with_origin else t


def _host_result_tensor(rows: int, cols: int):
    nbytes = rows * cols * 4
    if nbytes < (1 << 20) or _result_pool_cap() <= 0:
        return torch.empty((rows, cols), dtype=torch.float32), False
    import mmap
    with _RESULT_POOL_LOCK:
        mm = _RESULT_POOL.pop(nbytes, None)
        _RESULT_POOL.clear()         # another size: its pages go back to the system
    reused = mm is not None
    if mm is None:
        mm = mmap.mmap(-1, nbytes, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS, prot=mmap.PROT_READ | mmap.PROT_WRITE)
    flat = np.frombuffer(mm, dtype=np.float32)       # the tensor keeps `flat` alive (as the base of its array); it dies last
    weakref.finalize(flat, _give_back, mm, nbytes)
    return torch.from_numpy(flat.reshape(rows, cols)), reused


class HostAssembly:
    """``out[:, :f] = x`` (HOST, threads) started NOW, ``out[:, f:] = emb`` (DEVICE [N, K], DMA on the current stream) at
    :meth:`finish`, for a pageable HOST result `out` [N, f + K] (pope_assemble_begin / _finish): the page faults and the
    feature copy run underneath whatever happens between the two calls -- the upload of edge_index and the GPU work.
    Use as a context manager: an exception in between aborts the assembly (waits for the host threads)."""

    def __init__(self, x: torch.Tensor | None, out: torch.Tensor, f: int, threads: int = 0, chunks: int = 0):
        lib = _lib.load()
        assert not out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.dim() == 2 and out.shape[1] >= f
        assert f == 0 or (x is not None and not x.is_cuda and x.dtype == torch.float32 and x.shape == (out.shape[0], f) and x.stride(1) == 1)
        self.out, self.x, self.f, self.handle = out, x, f, None        # x is kept alive until finish
        if out.shape[0] == 0:
            return
        self.handle = lib.pope_assemble_begin(ptr(x) if f else None, (x.stride(0) * 4) if f else 0, f * 4, ptr(out), out.shape[1] * 4, out.shape[0],
                                              threads or host_threads(), chunks)
        if not self.handle:
            raise _lib.PopeError(_lib.ERR_INVALID, lib.pope_last_error().decode())

    def finish(self, emb: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        n, k = emb.shape
        assert emb.is_cuda and emb.is_contiguous() and emb.dtype == torch.float32 and self.out.shape == (n, self.f + k)
        h, self.handle = self.handle, None
        if h:
            with torch.cuda.device(emb.device):
                check(lib.pope_assemble_finish(h, ptr(emb), k * 4, k * 4, _stream()))
        return self.out

    def finish_codes(self, codes: torch.Tensor, lut: torch.Tensor) -> torch.Tensor:
        """:meth:`finish` for the embedding in its transport form (:func:`hop_codes`): uint8 [N, K] codes and the 256 floats
        they stand for, both on the device; a quarter of the bytes cross PCIe and the host threads look the floats up
        (pope_assemble_finish_codes; needs the process's pinned ring: pope_assemble_ring_ready)."""
        lib = _lib.load()
        n, k = codes.shape
        assert codes.is_cuda and codes.dtype == torch.uint8 and codes.stride(1) == 1 and self.out.shape == (n, self.f + k)
        assert lut.is_cuda and lut.dtype == torch.float32 and lut.numel() == 256 and lut.is_contiguous()
        h, self.handle = self.handle, None
        if h:
            with torch.cuda.device(codes.device):
                check(lib.pope_assemble_finish_codes(h, ptr(codes), codes.stride(0), k, ptr(lut), _stream()))
        return self.out

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.handle:
            _lib.load().pope_assemble_abort(self.handle)
            self.handle = None
        return False


def assemble_host_result(x: torch.Tensor | None, emb: torch.Tensor, out: torch.Tensor, f: int, threads: int = 0, chunks: int = 0) -> None:
    """out[:, :f] = x (HOST, threads) and out[:, f:] = emb (DEVICE [N, K], DMA on the current stream) for a pageable
    HOST result `out` [N, f + K]; returns when `out` is complete (pope_assemble_host_result)."""
    lib = _lib.load()
    assert not out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and emb.is_cuda and emb.stride(1) == 1
    n, k = emb.shape
    assert out.shape == (n, f + k) and (f == 0 or (x is not None and not x.is_cuda and x.shape == (n, f) and x.stride(1) == 1))
    with torch.cuda.device(emb.device):
        check(lib.pope_assemble_host_result(ptr(x) if f else None, (x.stride(0) * 4) if f else 0, f * 4, ptr(emb), emb.stride(0) * 4, k * 4,
                                            ptr(out), (f + k) * 4, n, threads or host_threads(), chunks, _stream()))


def copy_columns_to_host(src: torch.Tensor, dst: torch.Tensor) -> None:
    """dst (a HOST [N, K] column block of the result, rows contiguous, ideally pinned) = src (device [N, K], contiguous),
    asynchronously on the current stream (pope_copy_2d_to_host: one pitched DMA)."""
    lib = _lib.load()
    assert src.is_cuda and src.is_contiguous() and not dst.is_cuda and dst.shape == src.shape and dst.stride(1) == 1
    es = src.element_size()
    with torch.cuda.device(src.device):
        check(lib.pope_copy_2d_to_host(ptr(src), src.shape[1] * es, ptr(dst), dst.stride(0) * es, src.shape[1] * es,
                                       src.shape[0], _stream()))


class Csr:
    """Forward CSR on the device: slot p is the edge erow[p] -> col[p], slots sorted by erow."""

    def __init__(self, rowptr, col, erow, aux, num_nodes, num_edges, edge_index, checked):
        self.rowptr, self.col, self.erow, self.aux = rowptr, col, erow, aux     # int32 [N+1], [E], [E], [aux]
        self.num_nodes, self.num_edges = num_nodes, num_edges
        self.edge_index = edge_index        # kept so an unsorted edge list can be rebuilt on demand
        self.checked = checked              # False: built with defer_check, verdict still on the device


def build_csr(edge_index: torch.Tensor, num_nodes: int, defer_check: bool = False) -> Csr:
    """int64 [2, E] on the device -> Csr.  Replaces utils.py:121 ``to_networkx(data)``.

    ``defer_check=True`` skips the host synchronisation: index validation and the sortedness verdict are
    picked up by :func:`bfs`, which rebuilds through the general path if the edge list was not sorted.
    """
    lib = _lib.load()
    assert edge_index.is_cuda and edge_index.dtype == torch.int64 and edge_index.dim() == 2 and edge_index.shape[0] == 2
    ei = edge_index.contiguous()
    e = ei.shape[1]
    dev = ei.device
    with torch.cuda.device(dev):
        rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
        padded = (max(e, 1) + 63) // 64 * 64                      # the BFS kernel reads col / erow 16 bytes at a time
        col = torch.empty(padded, dtype=torch.int32, device=dev)
        erow = torch.empty(padded, dtype=torch.int32, device=dev)
        aux = torch.empty(lib.pope_csr_aux_elems(e), dtype=torch.int32, device=dev)
        scratch = _bytes(lib.pope_csr_scratch_bytes(num_nodes, e), dev)
        check(lib.pope_csr_build(ptr(ei), e, num_nodes, ptr(rowptr), ptr(col), ptr(erow), ptr(aux), ptr(scratch),
                                 scratch.numel(), 1 if defer_check else 0, _stream()))
    return Csr(rowptr, col, erow, aux, num_nodes, e, ei, not defer_check)


class HopPlanes:
    """Bit-sliced hop counts of K anchors (include/graphpope_hip.h, "hop planes")."""

    def __init__(self, planes: torch.Tensor, n_hop_bits: int, max_hop: int, num_nodes: int, k: int):
        self.planes = planes            # int64 [capacity + 1, N, W]; planes[:1 + n_hop_bits] valid
        self.n_hop_bits = n_hop_bits
        self.max_hop = max_hop
        self.num_nodes = num_nodes
        self.k = k

    def valid(self) -> torch.Tensor:
        return self.planes[: 1 + self.n_hop_bits]


MAX_CODED_HOP = 254                 # the byte code of hop h is h + 1; 0 says "no path"


def hop_codes(hp: HopPlanes):
    """The embedding of ``hp`` in its transport form: (uint8 [N, K] on the device -- 0 = no path, hops + 1 otherwise --,
    float32 [256] on the device: the value each code stands for).  Only for ``hp.max_hop <= MAX_CODED_HOP``
    (pope_geodesic_hop_codes)."""
    lib = _lib.load()
    dev = hp.planes.device
    assert hp.planes.is_contiguous() and hp.planes.shape[1] == hp.num_nodes
    with torch.cuda.device(dev):
        codes = torch.empty((hp.num_nodes, hp.k), dtype=torch.uint8, device=dev)
        lut = torch.empty(256, dtype=torch.float32, device=dev)
        check(lib.pope_geodesic_hop_codes(ptr(hp.planes), hp.n_hop_bits, hp.max_hop, hp.num_nodes, hp.k, ptr(codes), hp.k, ptr(lut), _stream()))
    return codes, lut


def bfs(csr: Csr, anchors, capacity: int = DEFAULT_PLANE_CAPACITY) -> HopPlanes:
    """Multi-source BFS for all anchors at once (utils.py:64-114).  Retries with more hop bits on overflow."""
    lib = _lib.load()
    anc = np.ascontiguousarray(np.asarray(anchors), dtype=np.int64)
    k = int(anc.size)
    dev = csr.col.device
    num_nodes = csr.num_nodes
    w = lib.pope_words(k)
    with torch.cuda.device(dev):
        scratch = _bytes(lib.pope_bfs_scratch_bytes(num_nodes, csr.num_edges, k), dev)
        while True:
            planes = torch.empty((capacity + 1, num_nodes, w), dtype=torch.int64, device=dev)
            max_hop, bits = ctypes.c_int32(0), ctypes.c_int32(0)
            rc = lib.pope_geodesic_bfs(ptr(csr.rowptr), ptr(csr.col), ptr(csr.erow), ptr(csr.aux), num_nodes, csr.num_edges,
                                       ctypes.c_void_p(anc.ctypes.data), k, ptr(planes), capacity, ptr(scratch),
                                       scratch.numel(), ctypes.byref(max_hop), ctypes.byref(bits), _stream())
            if rc == _lib.ERR_UNSORTED and not csr.checked:
                fixed = build_csr(csr.edge_index, num_nodes)          # general path (counting sort), synchronous
                csr.rowptr, csr.col, csr.erow, csr.aux, csr.checked = fixed.rowptr, fixed.col, fixed.erow, fixed.aux, True
                continue
            if rc == _lib.ERR_HOP_OVERFLOW and capacity < 31:
                capacity = min(31, capacity * 2)      # 8 -> 16 -> 31 hop bits
                continue
            check(rc)
            return HopPlanes(planes, int(bits.value), int(max_hop.value), num_nodes, k)


SPECULATIVE_HOP_BITS = 4          # hop-bit planes the library clears up front: hops < 16 are valid without knowing the depth
SPECULATIVE_LEVELS = 12           # levels pope_geodesic_bfs_begin enqueues (LEVEL_BATCH in csrc/geodesic.hip)


class PendingBfs:
    """A BFS whose first 12 levels are enqueued but not waited for (pope_geodesic_bfs_begin / _finish)."""

    def __init__(self, csr: Csr, anchors, capacity: int = DEFAULT_PLANE_CAPACITY):
        lib = _lib.load()
        self.csr, self.capacity = csr, capacity
        self.anc = np.ascontiguousarray(np.asarray(anchors), dtype=np.int64)
        self.k = int(self.anc.size)
        dev = csr.col.device
        w = lib.pope_words(self.k)
        with torch.cuda.device(dev):
            self.scratch = _bytes(lib.pope_bfs_scratch_bytes(csr.num_nodes, csr.num_edges, self.k), dev)
            self.planes = torch.empty((capacity + 1, csr.num_nodes, w), dtype=torch.int64, device=dev)
            check(lib.pope_geodesic_bfs_begin(*self._args(), _stream()))

    def _args(self):
        c = self.csr
        return (ptr(c.rowptr), ptr(c.col), ptr(c.erow), ptr(c.aux), c.num_nodes, c.num_edges,
                ctypes.c_void_p(self.anc.ctypes.data), self.k, ptr(self.planes), self.capacity, ptr(self.scratch),
                self.scratch.numel())

    def speculative_planes(self) -> torch.Tensor:
        """planes[0 : 1 + 4]: valid once finish() has reported n_hop_bits <= 4."""
        return self.planes[: 1 + SPECULATIVE_HOP_BITS]

    def verdict(self) -> torch.Tensor:
        """Device int32 [2], valid in stream order: (deepest level that reached something, CSR status flags).  The BFS is
        complete inside the speculative window iff flags == 0 and the level is < SPECULATIVE_LEVELS (then hops < 16:
        4 hop bits).  Lets the sharded path exchange the verdicts with the planes instead of synchronising here."""
        last_active = self.scratch[:4].view(torch.int32)             # BfsCtl.last_active heads the BFS scratch
        return torch.cat([last_active, self.csr.aux[2:3]])

    def finish(self):
        """Synchronise and return HopPlanes, or None if this needs the general path (deep graph, unsorted edges, overflow)."""
        lib = _lib.load()
        max_hop, bits = ctypes.c_int32(0), ctypes.c_int32(0)
        with torch.cuda.device(self.csr.col.device):
            rc = lib.pope_geodesic_bfs_finish(*self._args(), ctypes.byref(max_hop), ctypes.byref(bits), _stream())
        if rc in (_lib.ERR_UNSORTED, _lib.ERR_HOP_OVERFLOW):
            return None
        check(rc)
        return HopPlanes(self.planes, int(bits.value), int(max_hop.value), self.csr.num_nodes, self.k)


def finalize(planes: torch.Tensor, n_hop_bits: int, num_nodes: int, k: int, x, f: int, out: torch.Tensor, c0: int = 0):
    """Write x and 1/(hops+1) for one shard's k anchors into out[:, :f] and out[:, f+c0 : f+c0+k]."""
    lib = _lib.load()
    assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == num_nodes
    assert planes.is_contiguous() and planes.shape[0] >= 1 + n_hop_bits
    with torch.cuda.device(out.device):
        check(lib.pope_geodesic_finalize(ptr(planes), n_hop_bits, num_nodes, k, ptr(x), f, ptr(out), out.shape[1],
                                         c0, _stream()))


def copy_features(x: torch.Tensor, f: int, out: torch.Tensor):
    """out[:, :f] = x on the current stream (pope_concat): the feature half of the expansion on its own."""
    lib = _lib.load()
    if f == 0:
        return
    assert out.is_cuda and out.is_contiguous() and x.is_contiguous() and x.shape == (out.shape[0], f)
    with torch.cuda.device(out.device):
        check(lib.pope_concat(ptr(x), out.shape[0], f, ptr(out), out.shape[1], _stream()))


def finalize_shards(gathered: torch.Tensor, n_hop_bits: int, num_nodes: int, k_shard: int, x, f: int, out: torch.Tensor):
    """All shards of an all-gathered [world, 1 + bits, N, W] plane tensor in one pass over ``out``."""
    lib = _lib.load()
    assert gathered.is_contiguous() and gathered.dim() == 4 and gathered.shape[1] >= 1 + n_hop_bits
    world = gathered.shape[0]
    stride = gathered.shape[1] * gathered.shape[2] * gathered.shape[3]
    with torch.cuda.device(out.device):
        check(lib.pope_geodesic_finalize_shards(ptr(gathered), world, stride, n_hop_bits, num_nodes, k_shard, ptr(x), f,
                                                ptr(out), out.shape[1], _stream()))


def column_stats(hp: HopPlanes):
    """(hop_sum int64 [K], reach int64 [K]) on the device: per anchor, the sum of hop counts of the nodes that reach it
    and how many do (the anchor itself included)."""
    lib = _lib.load()
    dev = hp.planes.device
    with torch.cuda.device(dev):
        hop_sum = torch.empty(hp.k, dtype=torch.int64, device=dev)
        reach = torch.empty(hp.k, dtype=torch.int64, device=dev)
        scratch = _bytes(lib.pope_column_stats_scratch_bytes(hp.k), dev)
        check(lib.pope_geodesic_column_stats(ptr(hp.planes), hp.n_hop_bits, hp.num_nodes, hp.k, ptr(hop_sum), ptr(reach),
                                             ptr(scratch), scratch.numel(), _stream()))
    return hop_sum, reach


def closeness_centrality(edge_index: torch.Tensor, num_nodes: int, batch: int = 256) -> np.ndarray:
    """nx.closeness_centrality(to_networkx(data)) for every node, float64 [N], bit-identical to NetworkX 3.x.

    The multi-source BFS kernel is run with EVERY node as an anchor, `batch` at a time: hop(v -> a) is the inward
    distance NetworkX uses on a DiGraph.  The GPU returns exact integers (reach count, distance sum); the float formula
    ((r - 1) / totsp) * ((r - 1) / (n - 1)) (Wasserman-Faust scaling) is evaluated on the host in the same order.
    """
    dev = require_gpu(edge_index.device)
    csr = build_csr(edge_index.to(dev), num_nodes)
    sums = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    reach = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    for lo in range(0, num_nodes, batch):
        hi = min(lo + batch, num_nodes)
        hp = bfs(csr, np.arange(lo, hi))
        s, r = column_stats(hp)
        sums[lo:hi], reach[lo:hi] = s, r
    totsp = sums.cpu().numpy().astype(np.float64)
    r1 = reach.cpu().numpy().astype(np.float64) - 1.0
    out = np.zeros(num_nodes, dtype=np.float64)
    if num_nodes > 1:
        ok = totsp > 0.0
        out[ok] = (r1[ok] / totsp[ok]) * (r1[ok] / (num_nodes - 1))
    return out


def build_csr_canonical(edge_index: torch.Tensor, num_nodes: int) -> Csr:
    """Csr with every row's targets ascending (repeated edges adjacent), whatever the order of ``edge_index``."""
    lib = _lib.load()
    assert edge_index.is_cuda and edge_index.dtype == torch.int64 and edge_index.dim() == 2 and edge_index.shape[0] == 2
    ei = edge_index.contiguous()
    e, dev = ei.shape[1], ei.device
    with torch.cuda.device(dev):
        rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
        padded = (max(e, 1) + 63) // 64 * 64
        col = torch.empty(padded, dtype=torch.int32, device=dev)
        erow = torch.empty(padded, dtype=torch.int32, device=dev)
        aux = torch.empty(lib.pope_csr_aux_elems(e), dtype=torch.int32, device=dev)
        scratch = _bytes(lib.pope_csr_scratch_bytes(num_nodes, e), dev)
        check(lib.pope_csr_build_canonical(ptr(ei), e, num_nodes, ptr(rowptr), ptr(col), ptr(erow), ptr(aux), ptr(scratch),
                                           scratch.numel(), _stream()))
    return Csr(rowptr, col, erow, aux, num_nodes, e, ei, True)


def pagerank(edge_index: torch.Tensor, num_nodes: int, alpha: float = 0.85, max_iter: int = 100, tol: float = 1.0e-6) -> np.ndarray:
    """nx.pagerank(to_networkx(data)) (utils.py:26-30; NetworkX 3's SciPy power iteration) for every node, float64 [N],
    bit-identical to NetworkX: the sparse product runs on the GPU in SciPy's accumulation order (pope_pagerank_step); the
    two scalar reductions of an iteration -- the dangling mass (a Python sum in index order) and the l1 convergence norm
    (NumPy's pairwise sum) -- are evaluated here on the copied-back vector exactly as the reference's interpreter does.

    Raises RuntimeError if the iteration does not converge within ``max_iter`` (NetworkX: PowerIterationFailedConvergence)."""
    lib = _lib.load()
    dev = require_gpu(edge_index.device)
    n = int(num_nodes)
    if n == 0:
        return np.zeros(0, dtype=np.float64)
    ei = edge_index.to(dev, torch.int64).contiguous()
    by_source = build_csr_canonical(ei, n)
    by_target = build_csr_canonical(ei.flip(0).contiguous(), n)          # rows = targets, entries = sources, ascending
    with torch.cuda.device(dev):
        w = torch.empty(n, dtype=torch.float64, device=dev)
        check(lib.pope_pagerank_weights(ptr(by_source.rowptr), ptr(by_source.col), n, ptr(w), _stream()))
        dangling = np.where(w.cpu().numpy() == 0.0)[0]
        x_host = np.repeat(1.0 / n, n)
        bufs = [torch.as_tensor(x_host, device=dev), torch.empty(n, dtype=torch.float64, device=dev)]
        staging = torch.empty(n, dtype=torch.float64, pin_memory=True)
        for it in range(max_iter):
            dsum = float(sum(x_host[dangling]))                            # the reference's Python sum, index order
            check(lib.pope_pagerank_step(ptr(by_target.rowptr), ptr(by_target.col), n, ptr(bufs[0]), ptr(w), dsum, float(alpha),
                                         ptr(bufs[1]), _stream()))
            staging.copy_(bufs[1], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            x_new = staging.numpy().copy()
            err = np.absolute(x_new - x_host).sum()
            x_host = x_new
            bufs.reverse()
            if err < n * tol:
                return x_host
    raise RuntimeError(f"pagerank: power iteration failed to converge within {max_iter} iterations")


def hop_matrix(hp: HopPlanes) -> torch.Tensor:
    """int32 [N, K], -1 = unreachable (the integers behind the reference's floats)."""
    lib = _lib.load()
    dev = hp.planes.device
    with torch.cuda.device(dev):
        hops = torch.empty((hp.num_nodes, hp.k), dtype=torch.int32, device=dev)
        check(lib.pope_geodesic_hops(ptr(hp.planes), hp.n_hop_bits, hp.num_nodes, hp.k, ptr(hops), _stream()))
    return hops


_WORKSPACE = {}          # (device, stream, bytes) -> uint8 tensor, reused by geodesic_run(reuse_workspace=True)


def geodesic_run(x, edge_index: torch.Tensor, num_nodes: int, anchors, capacity: int = DEFAULT_PLANE_CAPACITY,
                 want_out: bool = True, reuse_workspace: bool = False):
    """The whole geodesic hot path in one library call (one host synchronisation): (out, HopPlanes).

    ``x`` float32 [N, F] on the device, or None: then ``out`` is the [N, K] embedding alone (``want_out=False``: BFS only).
    ``reuse_workspace`` keeps the scratch allocation between calls; the returned HopPlanes then only stay valid
    until the next such call on the same stream.
    """
    lib = _lib.load()
    ei = edge_index.contiguous()
    dev = ei.device
    assert ei.is_cuda and ei.dtype == torch.int64 and ei.dim() == 2 and ei.shape[0] == 2
    anc = np.ascontiguousarray(np.asarray(anchors), dtype=np.int64)
    k, e = int(anc.size), ei.shape[1]
    f = 0 if x is None else x.shape[1]
    w = lib.pope_words(k)
    trace = os.environ.get("GRAPHPOPE_TRACE")
    t_in = time.perf_counter()
    with torch.cuda.device(dev):
        out = torch.empty((num_nodes, f + k), dtype=torch.float32, device=dev) if want_out else None
        while True:
            nbytes_ws = lib.pope_geodesic_run_workspace_bytes(num_nodes, e, k, capacity)
            if reuse_workspace:
                # keyed by the launch stream as well: the finalise kernel of one call may still be reading the hop
                # planes when the call returns, and only work on the SAME stream is ordered behind it
                key = (dev, torch.cuda.current_stream().cuda_stream, nbytes_ws)
                ws = _WORKSPACE.get(key)
                if ws is None:
                    for old in [k for k in _WORKSPACE if k[:2] == key[:2]]:
                        del _WORKSPACE[old]
                    ws = _WORKSPACE.setdefault(key, _bytes(nbytes_ws, dev))
            else:
                ws = _bytes(nbytes_ws, dev)
            max_hop, bits = ctypes.c_int32(0), ctypes.c_int32(0)
            t_call = time.perf_counter()
            rc = lib.pope_geodesic_run(ptr(ei), e, num_nodes, ctypes.c_void_p(anc.ctypes.data), k, ptr(x), f, ptr(out),
                                       f + k, capacity, ptr(ws), ws.numel(), ctypes.byref(max_hop), ctypes.byref(bits),
                                       _stream())
            if trace:
                print(f"[trace] geodesic_run: allocations {1e3 * (t_call - t_in):.2f} ms, pope_geodesic_run {1e3 * (time.perf_counter() - t_call):.2f} ms",
                      file=sys.stderr)
            if rc == _lib.ERR_HOP_OVERFLOW and capacity < 31:
                capacity = min(31, capacity * 2)
                continue
            check(rc)
            off = lib.pope_geodesic_run_planes(ptr(ws), num_nodes, e, k, capacity) - ws.data_ptr()
            nbytes = (capacity + 1) * num_nodes * w * 8
            planes = ws[off: off + nbytes].view(torch.int64).view(capacity + 1, num_nodes, w)
            return out, HopPlanes(planes, int(bits.value), int(max_hop.value), num_nodes, k)


def geodesic_features(x: torch.Tensor, edge_index: torch.Tensor, num_nodes: int, anchors, group=None,
                      shard: bool = True) -> torch.Tensor:
    """[N, F+K] float32 on the device: features next to the geodesic POPE embedding (utils.py:137-147).
    ``x=None``: the [N, K] embedding alone (the host -> host caller keeps the features on the host).

    With an initialised ``torch.distributed`` group of more than one rank the anchors are sharded
    over the ranks and the hop planes are all-gathered; every rank returns the full matrix.
    ``shard=False`` computes all anchors locally even inside a process group.
    """
    from . import distributed as pdist
    dev = require_gpu(edge_index.device if x is None else x.device)
    anc = np.asarray(anchors, dtype=np.int64)
    world = pdist.world_size(group) if shard else 1
    if x is not None:
        assert x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] == num_nodes
        x = x.contiguous()
    if world == 1:
        return geodesic_run(x, edge_index.to(dev), num_nodes, anc, reuse_workspace=True)[0]
    if x is None:
        x = torch.empty((num_nodes, 0), dtype=torch.float32, device=dev)
    csr = build_csr(edge_index.to(dev), num_nodes, defer_check=True)
    return pdist.sharded_geodesic_features(
        x, num_nodes, anc, group,
        bfs_fn=lambda a: bfs(csr, a),
        finalize_fn=finalize, finalize_all_fn=finalize_shards, begin_fn=lambda a: PendingBfs(csr, a),
        copy_x_fn=copy_features)


# ------------------------------------------------------------------------------------------------
# node2vec-space embedding (utils.py:149-180)
# ------------------------------------------------------------------------------------------------
def _anchor_ids(anchors, n: int, dev) -> torch.Tensor:
    """The anchor row ids as a device int64 tensor.  Ids given on the host are range-checked here, as indexing would (the
    kernel follows them without a check); ids that already live on the device are the caller's responsibility, like any
    index tensor handed to a kernel -- checking them would cost a synchronisation."""
    if isinstance(anchors, torch.Tensor) and anchors.device == dev and anchors.dtype == torch.int64:
        return anchors.contiguous()
    arr = np.ascontiguousarray(anchors.cpu().numpy() if isinstance(anchors, torch.Tensor) else np.asarray(anchors), dtype=np.int64).reshape(-1)
    if arr.size and (int(arr.min()) < -n or int(arr.max()) >= n):
        raise IndexError(f"anchor id out of range for a table of {n} rows")
    arr = np.where(arr < 0, arr + n, arr)                # numpy / torch indexing semantics for negative ids
    return torch.as_tensor(arr, device=dev)


def pairwise_embedding(emb: torch.Tensor, anchors, distance_function: str, anchor_embeddings=None, out: torch.Tensor = None,
                       c0: int = 0) -> torch.Tensor:
    """Min-max scaled distance of every node2vec row to the anchor rows (``anchors`` = row indices, utils.py:165-167) or
    to ``anchor_embeddings`` [K, D] (K-means centres, utils.py:168-170), float32 on the device: a new [N, K] matrix, or
    columns [c0, c0 + K) of ``out``."""
    lib = _lib.load()
    dev = require_gpu(emb.device)
    metric = _lib.METRIC[distance_function]          # KeyError for unknown names, as utils.py:164
    emb = emb.to(dev, torch.float32).contiguous()
    n, d = emb.shape
    if anchor_embeddings is not None:
        a = torch.as_tensor(anchor_embeddings).to(dev, torch.float32).contiguous()
        assert a.dim() == 2 and a.shape[1] == d
    else:                                                # the K anchor rows (utils.py:167) are read through their ids
        a = None
        idx = _anchor_ids(anchors, n, dev)
    k = a.shape[0] if a is not None else idx.numel()
    with torch.cuda.device(dev):
        if out is None:
            out = torch.empty((n, k), dtype=torch.float32, device=dev)
        assert out.is_cuda and out.is_contiguous() and out.shape[0] == n and out.shape[1] >= c0 + k
        if a is not None:
            scratch = _bytes(lib.pope_pairwise_scratch_bytes(n, k, d), dev)
            check(lib.pope_pairwise_minmax(ptr(emb), n, d, ptr(a), k, metric, ptr(out), out.shape[1], c0, ptr(scratch),
                                           scratch.numel(), _stream()))
        else:
            scratch = _bytes(lib.pope_pairwise_by_id_scratch_bytes(n, k, d), dev)
            check(lib.pope_pairwise_features_by_id(None, 0, ptr(emb), n, d, ptr(idx), k, metric, ptr(out), out.shape[1], c0, ptr(scratch),
                                                   scratch.numel(), _stream()))
    return out


def _relocate_empty_clusters(xc: torch.Tensor, centers_old: torch.Tensor, centers_new: torch.Tensor, labels: torch.Tensor, k: int) -> bool:
    """scikit-learn's ``_relocate_empty_clusters_dense`` (sklearn/cluster/_k_means_common.pyx) on the result of one Lloyd
    update: every empty cluster takes, in order, one of the points that lie farthest from their own (old) centre -- the
    farthest point goes to the first empty cluster -- and that point leaves the cluster it was counted in.  `centers_new`
    holds MEANS (pope_kmeans_lloyd_step keeps an empty cluster's old centre), so the donor's mean is rebuilt from its sum.
    Empty clusters are rare (k-means++ seeds every cluster with a data point): a handful of torch ops, run only then.
    Returns True if anything was moved."""
    counts = torch.bincount(labels.to(torch.int64), minlength=k)
    empty = torch.nonzero(counts == 0, as_tuple=False).flatten()
    n_empty = int(empty.numel())
    if n_empty == 0:
        return False
    lab = labels.to(torch.int64)
    dist = ((xc - centers_old.index_select(0, lab)).double() ** 2).sum(1)
    far = torch.argsort(dist, descending=True, stable=True)[:n_empty]            # np.argpartition(...)[:-n_empty-1:-1]: farthest first
    sums = centers_new.double() * counts.clamp(min=1).double()[:, None]
    cnt = counts.clone().double()
    for new_id, idx in zip(empty.tolist(), far.tolist()):
        old_id = int(lab[idx])
        sums[old_id] -= xc[idx].double()
        sums[new_id] = xc[idx].double()
        cnt[new_id] = 1.0
        cnt[old_id] -= 1.0
    fixed = (sums / cnt.clamp(min=1.0)[:, None]).to(torch.float32)
    touched = torch.unique(torch.cat([empty, lab[far]]))
    centers_new[touched] = fixed[touched]
    return True


def kmeans_centers(emb: torch.Tensor, n_clusters: int, max_iter: int = 300, tol: float = 1e-4):
    """``KMeans(n_clusters=K).fit(X).cluster_centers_`` (utils.py:168-170, scikit-learn defaults: k-means++ seeding with
    2 + int(log K) local trials, one run, Lloyd on the mean-centred data) with the distances, reductions and centre updates on
    the GPU.  The random numbers are drawn HERE from the global legacy NumPy stream, call for call as scikit-learn's
    ``check_random_state(None)`` would (one ``choice`` for the first seed, ``uniform(size=trials)`` per further seed), so the
    stream is left where the reference leaves it.  float32 [K, D] on the device.

    Not bit-identical to scikit-learn (whose own result depends on its BLAS chunking); on well-separated data the same
    points are seeded in the same order and the centres agree to float32 rounding.  On overlapping data (an untrained
    node2vec table) a rounding difference in the seeding's cumulative sums can pick another seed and the centres then differ
    as two runs of scikit-learn with different seeds do: parity in distribution only, which is why ``utils.attach_node2vec``
    makes the reference's own scikit-learn call unless GRAPHPOPE_KMEANS=gpu asks for this one."""
    lib = _lib.load()
    dev = require_gpu(emb.device)
    x = emb.to(dev, torch.float32).contiguous()
    n, d = x.shape
    k = int(n_clusters)
    if not 0 < k <= n:
        raise ValueError(f"n_samples={n} should be >= n_clusters={k}.")           # scikit-learn's message
    with torch.cuda.device(dev):
        scratch = _bytes(max(lib.pope_kmeans_scratch_bytes(n, d, k), 4096 * d * 8), dev)
        sums = torch.empty((2, d), dtype=torch.float64, device=dev)
        check(lib.pope_column_moments(ptr(x), n, d, ptr(sums[0]), ptr(sums[1]), ptr(scratch), scratch.numel(), _stream()))
        s = sums.cpu().numpy()
        mean64 = s[0] / n
        tol_abs = float(np.mean(s[1] / n - mean64 * mean64) * tol)          # _tolerance: mean column variance x tol
        mean = torch.as_tensor(mean64.astype(np.float32), device=dev)
        xc = torch.empty_like(x)
        check(lib.pope_shift_columns(ptr(x), ptr(mean), n, d, -1.0, ptr(xc), _stream()))      # KMeans.fit: X -= X.mean(axis=0)
        # k-means++ seeding (_kmeans_plusplus): the global stream, in scikit-learn's order
        trials = 2 + int(np.log(k))
        weights = np.ones(n, dtype=np.float32)
        first = int(np.random.choice(n, p=weights / weights.sum()))
        uniforms = np.ascontiguousarray(np.stack([np.random.uniform(size=trials) for _ in range(k - 1)]) if k > 1
                                        else np.zeros((0, trials)), dtype=np.float64)
        chosen = torch.empty(k, dtype=torch.int64, device=dev)
        check(lib.pope_kmeans_plusplus(ptr(xc), n, d, k, first, ctypes.c_void_p(uniforms.ctypes.data), trials, ptr(chosen),
                                       ptr(scratch), scratch.numel(), _stream()))
        centers = xc.index_select(0, chosen).contiguous()
        centers_new = torch.empty_like(centers)
        labels = torch.full((n,), -1, dtype=torch.int32, device=dev)
        labels_prev = torch.full((n,), -1, dtype=torch.int32, device=dev)
        changed = torch.zeros(1, dtype=torch.int32, device=dev)
        shift = torch.zeros(1, dtype=torch.float64, device=dev)
        for _ in range(max_iter):                                             # _kmeans_single_lloyd
            check(lib.pope_kmeans_lloyd_step(ptr(xc), n, d, ptr(centers), k, ptr(centers_new), ptr(labels), ptr(labels_prev),
                                             ptr(changed), ptr(shift), ptr(scratch), scratch.numel(), _stream()))
            relocated = _relocate_empty_clusters(xc, centers, centers_new, labels, k)   # rare: scikit-learn's rule for empty clusters
            centers, centers_new = centers_new, centers
            labels, labels_prev = labels_prev, labels                          # labels_prev now holds this iteration's labels
            if int(changed.item()) == 0:                                      # strict convergence: the labels did not move
                break
            total_shift = float(((centers - centers_new).double() ** 2).sum().item()) if relocated else float(shift.item())
            if total_shift <= tol_abs:
                break
        out = torch.empty_like(centers)
        check(lib.pope_shift_columns(ptr(centers), ptr(mean), k, d, 1.0, ptr(out), _stream()))   # best_centers += X_mean
    return out


def pairwise_features(x: torch.Tensor, emb: torch.Tensor, anchors, distance_function: str,
                      anchor_embeddings=None) -> torch.Tensor:
    """[N, F+K] float32 on the device: the features next to :func:`pairwise_embedding` (device-resident callers); the
    feature copy rides inside the tile kernel (pope_pairwise_features)."""
    lib = _lib.load()
    dev = require_gpu(x.device)
    metric = _lib.METRIC[distance_function]
    x = x.contiguous()
    emb = emb.to(dev, torch.float32).contiguous()
    n, f = x.shape
    d = emb.shape[1]
    if anchor_embeddings is not None:
        a = torch.as_tensor(anchor_embeddings).to(dev, torch.float32).contiguous()
        assert a.dim() == 2 and a.shape[1] == d
    else:                                                                 # embedding[anchor_nodes] (utils.py:167): read through the ids
        a = None
        idx = _anchor_ids(anchors, emb.shape[0], dev)
    k = a.shape[0] if a is not None else idx.numel()
    with torch.cuda.device(dev):
        out = torch.empty((n, f + k), dtype=torch.float32, device=dev)
        if a is not None:
            scratch = _bytes(lib.pope_pairwise_scratch_bytes(n, k, d), dev)
            check(lib.pope_pairwise_features(ptr(x), f, ptr(emb), n, d, ptr(a), k, metric, ptr(out), f + k, f, ptr(scratch),
                                             scratch.numel(), _stream()))
        else:
            scratch = _bytes(lib.pope_pairwise_by_id_scratch_bytes(n, k, d), dev)
            check(lib.pope_pairwise_features_by_id(ptr(x), f, ptr(emb), n, d, ptr(idx), k, metric, ptr(out), f + k, f, ptr(scratch),
                                                   scratch.numel(), _stream()))
    return out
