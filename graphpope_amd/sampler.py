"""Device-side mini-batch sampling: the GPU stand-in for PyG ``NeighborSampler`` (main.py:100-116) and
``convert_batch`` (main.py:118-123).  SURVEY.md §8f rank 1.

    sampler = NeighborSampler(rowptr, col, num_nodes, sizes=[25, 10])       # CSR on the device (engine.build_csr)
    n_id, adjs = sampler.sample(seeds, seed=epoch_seed)                      # adjs: outer -> inner, like the reference
    x = data_x_device.index_select(0, n_id)                                  # features stay resident in HBM

``sizes`` has PyG's meaning: the first entry is the fan-out around the seed nodes, and the returned list is reversed
(outermost block first), so ``model(x, adjs)`` consumes it exactly like the reference's ``Batch.adjs_t``.
The draw is a pure function of (seed, hop, node): reproducible, and checked bit for bit against a CPU restatement.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check, on_device, ptr
from .sage import SampledAdj


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceBatch:
    """Fixed buffers for mini-batches of `n_seeds` seeds and fan-outs `sizes`, sized by CAPACITY (hop h may meet up to
    t_cap[h] = t_cap[h-1] * (1 + sizes[h-1]) targets): the device-extent sampler fills them without ever telling the host
    how large the batch came out -- ``dims[h]`` = {n_dst, n_src, nnz, 0} stays on the device and every SAGE kernel reads
    its sizes there.  The same buffers are filled again for the next batch (a captured HIP graph replays on them)."""

    def __init__(self, n_seeds: int, sizes, device):
        self.n_seeds, self.sizes = int(n_seeds), [int(f) for f in sizes]
        assert all(f > 0 for f in self.sizes), "device-extent batches need positive fan-outs"
        h = len(self.sizes)
        self.t_cap, self.caps = [], []
        t = self.n_seeds
        for f in self.sizes:
            self.t_cap.append(t)
            self.caps.append(t * f)
            t = t + t * f
        self.rowptrs = [torch.zeros(tc + 1, dtype=torch.int32, device=device) for tc in self.t_cap]
        self.cols = [torch.zeros(max(c, 1), dtype=torch.int32, device=device) for c in self.caps]
        self.n_ids = [torch.zeros(tc + c, dtype=torch.int64, device=device) for tc, c in zip(self.t_cap, self.caps)]
        self.dims = torch.zeros((h, 4), dtype=torch.int32, device=device)
        self.n_id = self.n_ids[-1]                                   # [capacity]: the true length is dims[-1, 1]
        self.host_dims = None                                        # set by load(): sizes known on the host
        self._sampler_scratch = {}                                   # bytes -> tensor, see sampler_scratch()
        # outermost block first, like NeighborSampler's adjs (main.py:118-123)
        self.adjs = [SampledAdj(self.rowptrs[i], self.cols[i], self.t_cap[i] + self.caps[i], self.dims[i]) for i in range(h)][::-1]

    def sampler_scratch(self, nbytes: int) -> torch.Tensor:
        """The device-extent sampler's scratch for THIS batch (position-key map, ranks, chain status words): allocated once per
        size and kept for the life of the batch.  A captured step has its address baked in, so it must never be replaced or
        shared with the host-sized sampling paths (which regrow their own scratch on demand, NeighborSampler._scratch)."""
        t = self._sampler_scratch.get(nbytes)
        if t is None:
            t = self._sampler_scratch[nbytes] = torch.empty(nbytes, dtype=torch.uint8, device=self.dims.device)
        return t

    def segments(self, valid_only: bool = False):
        """Every buffer a batch consists of (same order for every DeviceBatch of the same shape): for sage_copy_segments.
        valid_only: the prefixes a batch loaded with :meth:`load` really fills (its sizes are known on the host) -- a pooled
        batch is then moved with ~1 MB instead of the 5.5 MB of capacity."""
        if valid_only and self.host_dims is not None:
            d = self.host_dims
            return ([r[: d[i][0] + 1] for i, r in enumerate(self.rowptrs)] + [c[: max(d[i][2], 1)] for i, c in enumerate(self.cols)]
                    + [self.n_id[: d[-1][1]], self.dims])
        return self.rowptrs + self.cols + [self.n_id, self.dims]

    def load(self, n_id: torch.Tensor, adjs) -> None:
        """Fill the buffers from a batch sampled elsewhere (host-sized SampledAdjs, outermost first): plain copies; the
        sizes travel as four device words per hop."""
        h = len(self.sizes)
        assert len(adjs) == h
        self.n_id[: n_id.numel()].copy_(n_id)
        host_dims = []
        for i, adj in enumerate(adjs[::-1]):                       # hop order
            self.rowptrs[i][: adj.n_dst + 1].copy_(adj.rowptr)
            self.cols[i][: adj.col.numel()].copy_(adj.col)
            host_dims.append([adj.n_dst, adj.n_src, int(adj.col.numel()), 0])
        self.dims.copy_(torch.tensor(host_dims, dtype=torch.int32))
        self.host_dims = host_dims


class NeighborSampler:
    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, num_nodes: int, sizes=(25, 10)):
        assert rowptr.is_cuda and rowptr.dtype == torch.int32 and col.dtype == torch.int32
        self.rowptr, self.col, self.num_nodes, self.sizes = rowptr, col, int(num_nodes), list(sizes)
        self._scratch = None

    def sample_device(self, seeds: torch.Tensor, seed: int = 0, out: DeviceBatch | None = None,
                      seed_dev: torch.Tensor | None = None) -> DeviceBatch:
        """The batch around `seeds` (int64 on the device, out.n_seeds of them) WITHOUT any host synchronisation
        (sage_sample_batch_device): `out`'s buffers are filled, the sizes stay in out.dims.  The draw equals
        ``sample(seeds, seed + int(seed_dev))``; capturable into a HIP graph (seed_dev: device int64 scalar)."""
        lib = _lib.load()
        dev = self.rowptr.device
        if out is None:
            out = DeviceBatch(seeds.numel(), self.sizes, dev)
        assert seeds.is_cuda and seeds.dtype == torch.int64 and seeds.is_contiguous() and seeds.numel() == out.n_seeds
        assert out.sizes == [int(f) for f in self.sizes]
        h = len(out.sizes)
        scratch = out.sampler_scratch(lib.sage_sample_scratch_bytes(self.num_nodes, out.t_cap[-1], out.caps[-1]))
        arr = ctypes.c_void_p * h
        with on_device(dev):
            check(lib.sage_sample_batch_device(ptr(self.rowptr), ptr(self.col), self.num_nodes, ptr(seeds), seeds.numel(),
                                               (ctypes.c_int32 * h)(*out.sizes), h, int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(seed_dev),
                                               arr(*[r.data_ptr() for r in out.rowptrs]), arr(*[c.data_ptr() for c in out.cols]),
                                               arr(*[n.data_ptr() for n in out.n_ids]), ptr(out.dims), ptr(scratch),
                                               scratch.numel(), _stream()))
        return out

    def sample_epoch_device(self, order: torch.Tensor, first_dev: torch.Tensor, labels: torch.Tensor | None, y_out: torch.Tensor | None,
                            seed: int = 0, out: DeviceBatch | None = None, seed_dev: torch.Tensor | None = None) -> DeviceBatch:
        """:meth:`sample_device` around the seeds ``order[first : first + out.n_seeds]`` with ``first`` read from the device word
        `first_dev` (int64), and -- with `labels` -- ``y_out[:] = labels[seeds]`` written by the sampler's first kernel
        (sage_sample_epoch_batch_device; main.py:100-123 without a loader).  No host synchronisation; capturable."""
        lib = _lib.load()
        dev = self.rowptr.device
        assert out is not None and order.is_cuda and order.dtype == torch.int64 and order.is_contiguous()
        assert first_dev.is_cuda and first_dev.dtype == torch.int64 and first_dev.numel() == 1
        assert (labels is None) == (y_out is None)
        if labels is not None:
            assert labels.is_cuda and labels.dtype == torch.int64 and labels.is_contiguous() and y_out.dtype == torch.int64
            assert y_out.is_contiguous() and y_out.numel() >= out.n_seeds
        h = len(out.sizes)
        scratch = out.sampler_scratch(lib.sage_sample_scratch_bytes(self.num_nodes, out.t_cap[-1], out.caps[-1]))
        arr = ctypes.c_void_p * h
        with on_device(dev):
            check(lib.sage_sample_epoch_batch_device(ptr(self.rowptr), ptr(self.col), self.num_nodes, ptr(order), ptr(first_dev), out.n_seeds,
                                                     ptr(labels), ptr(y_out), (ctypes.c_int32 * h)(*out.sizes), h,
                                                     int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(seed_dev),
                                                     arr(*[r.data_ptr() for r in out.rowptrs]), arr(*[c.data_ptr() for c in out.cols]),
                                                     arr(*[n.data_ptr() for n in out.n_ids]), ptr(out.dims), ptr(scratch),
                                                     scratch.numel(), _stream()))
        return out

    def _hop(self, targets: torch.Tensor, fanout: int, seed: int, hop: int):
        lib = _lib.load()
        dev = targets.device
        t = targets.numel()
        if fanout < 0:                                     # keep every neighbour: the edge count is the sum of the degrees
            deg = (self.rowptr[1:] - self.rowptr[:-1]).index_select(0, targets)
            cap = int(deg.sum().item())
        else:
            cap = t * fanout
        need = lib.sage_sample_scratch_bytes(self.num_nodes, t, cap)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=dev)
        out_rowptr = torch.empty(t + 1, dtype=torch.int32, device=dev)
        out_col = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
        n_id = torch.empty(t + cap, dtype=torch.int64, device=dev)
        nnz, n_src = ctypes.c_int64(0), ctypes.c_int64(0)
        with on_device(dev):
            check(lib.sage_sample_hop(ptr(self.rowptr), ptr(self.col), self.num_nodes, ptr(targets), t, fanout, seed, hop,
                                      ptr(out_rowptr), ptr(out_col), cap, ptr(n_id), ctypes.byref(nnz), ctypes.byref(n_src),
                                      ptr(self._scratch), self._scratch.numel(), _stream()))
        return n_id[: n_src.value], SampledAdj(out_rowptr, out_col[: nnz.value], n_src.value)

    def sample(self, seeds: torch.Tensor, seed: int = 0):
        """(n_id int64 on the device, [SampledAdj outer ... inner])."""
        n_id = seeds.to(self.rowptr.device, torch.int64).contiguous()
        if all(int(f) > 0 for f in self.sizes):
            return self._sample_batch(n_id, int(seed) & 0xFFFFFFFFFFFFFFFF)
        adjs = []
        for hop, size in enumerate(self.sizes):                    # 'all neighbours' hops: sized hop by hop
            n_id, adj = self._hop(n_id, int(size), int(seed) & 0xFFFFFFFFFFFFFFFF, hop)
            adjs.append(adj)
        return n_id, adjs[::-1]

    def _sample_batch(self, seeds: torch.Tensor, seed: int):
        """Every hop in ONE library call (sage_sample_batch): buffers sized by capacity, sliced to the counts it returns."""
        lib = _lib.load()
        dev = seeds.device
        h = len(self.sizes)
        t_cap, caps = [], []
        t = seeds.numel()
        for f in self.sizes:
            t_cap.append(t)
            caps.append(t * int(f))
            t = t + t * int(f)
        need = lib.sage_sample_scratch_bytes(self.num_nodes, t_cap[-1], caps[-1])
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=dev)
        rowptrs = [torch.empty(tc + 1, dtype=torch.int32, device=dev) for tc in t_cap]
        cols = [torch.empty(max(c, 1), dtype=torch.int32, device=dev) for c in caps]
        n_ids = [torch.empty(tc + c, dtype=torch.int64, device=dev) for tc, c in zip(t_cap, caps)]
        arr = ctypes.c_void_p * h
        nnz, n_src = (ctypes.c_int64 * h)(), (ctypes.c_int64 * h)()
        with on_device(dev):
            check(lib.sage_sample_batch(ptr(self.rowptr), ptr(self.col), self.num_nodes, ptr(seeds), seeds.numel(),
                                        (ctypes.c_int32 * h)(*[int(f) for f in self.sizes]), h, seed,
                                        arr(*[r.data_ptr() for r in rowptrs]), arr(*[c.data_ptr() for c in cols]),
                                        arr(*[n.data_ptr() for n in n_ids]), nnz, n_src, ptr(self._scratch),
                                        self._scratch.numel(), _stream()))
        adjs, t = [], seeds.numel()
        for i in range(h):
            adjs.append(SampledAdj(rowptrs[i][: t + 1], cols[i][: nnz[i]], n_src[i]))
            t = n_src[i]
        return n_ids[-1][: n_src[h - 1]], adjs[::-1]
