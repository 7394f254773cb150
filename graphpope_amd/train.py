"""One GraphSAGE training step with no host in the loop: sampled, run and optimised entirely by kernels that read their
sizes on the device, captured once into a HIP graph and replayed.

What the reference does per step (/root/reference/main.py): a DataLoader worker samples the batch
(NeighborSampler(sizes=[25, 10]), main.py:100-116) and gathers ``data.x[n_id]`` (convert_batch, main.py:118-123);
Lightning moves it to the GPU, calls ``training_step`` (main.py:213-222: forward + ``F.cross_entropy``), backward,
clips the gradient norm to 0.5 (main.py:286) and steps Adam (main.py:244).  Here the same step is

    sample (device-extent sampler, no read-back)  ->  SAGE forward on IndexedFeatures  ->  cross-entropy
    ->  backward  ->  [clip]  ->  one-launch Adam  ->  advance the device-side seeds / step count

enqueued once through the ordinary autograd path while a HIP graph is being captured, then replayed: a step costs one
small launch that loads the seeds and labels plus one graph launch.  Batches come either from the device sampler inside
the graph (`sampler` given) or from a pool of pre-sampled batches loaded into the graph's fixed buffers
(:meth:`load_batch`).  The data-dependent sizes of a batch never reach the host (include/graphpope_hip.h, "Device
extents"), so nothing in the step synchronises.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check, on_device
from .sage import IndexedFeatures, cross_entropy
from .sampler import DeviceBatch

_SEED_STRIDE = 0x9E3779B97F4A7C15 % (1 << 63)        # odd: the seed words walk through all 2^64 values


class StepState:
    """Four device words a replayed step reads and one launch advances: dropout seed, sampling seed, Adam step count, and
    the position in the epoch's node order (epoch mode: the next batch's seeds start there)."""

    def __init__(self, device, seed: int = 0, adam_step: int = 0):
        self.words = torch.tensor([seed, seed ^ 0x5DEECE66D, adam_step + 1, 0], dtype=torch.int64, device=device)
        self._inc = (ctypes.c_int64 * 4)(_SEED_STRIDE, _SEED_STRIDE | 2, 1, 0)
        self._inc_keep_sample = (ctypes.c_int64 * 4)(_SEED_STRIDE, 0, 1, 0)
        self._inc_only_sample = (ctypes.c_int64 * 4)(0, _SEED_STRIDE | 2, 0, 0)

    dropout_seed = property(lambda self: self.words[0:1])
    sample_seed = property(lambda self: self.words[1:2])
    adam_step = property(lambda self: self.words[2:3])
    cursor = property(lambda self: self.words[3:4])

    def advance(self, sample_seed: bool = True, cursor_by: int = 0) -> None:
        lib = _lib.load()
        inc = self._inc if sample_seed else self._inc_keep_sample
        if cursor_by:
            inc = (ctypes.c_int64 * 4)(inc[0], inc[1], inc[2], cursor_by)
        with on_device(self.words.device):
            check(lib.sage_advance_counters(ctypes.c_void_p(self.words.data_ptr()), inc, 4,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def advance_sample_seed(self) -> None:
        """Only the sampling seed (a batch sampled ahead on a side stream advances it there, in sampling order)."""
        lib = _lib.load()
        with on_device(self.words.device):
            check(lib.sage_advance_counters(ctypes.c_void_p(self.words.data_ptr()), self._inc_only_sample, 4,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


def copy_segments(dst, src) -> None:
    """dst[i][:] = src[i] for lists of contiguous device tensors (src[i] may be shorter than dst[i]) in ONE launch."""
    lib = _lib.load()
    n = len(dst)
    assert n == len(src)
    arr = ctypes.c_void_p * n
    nbytes = []
    for d, s in zip(dst, src):
        assert d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype
        assert s.numel() <= d.numel()
        nbytes.append(s.numel() * s.element_size())
    with on_device(dst[0].device):
        check(lib.sage_copy_segments(n, arr(*[d.data_ptr() for d in dst]), arr(*[s.data_ptr() for s in src]),
                                     (ctypes.c_int64 * n)(*nbytes), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))


class SageTrainStep:
    """``step(seeds, y)`` = one optimiser step of `model` on the batch around `seeds`; see the module docstring.

    model      graphpope_amd.sage.SAGE (training mode is set here)
    opt        graphpope_amd.optim.Adam over model.parameters()
    feats      float32 [N, C] on the device: features (+) POPE, resident in HBM
    sampler    graphpope_amd.sampler.NeighborSampler, or None: batches are then loaded with :meth:`load_batch`
    clip       max gradient norm (Lightning's gradient_clip_val), or None
    graph      False: the same step, enqueued eagerly every time (A/B and debugging)
    """

    def __init__(self, model, opt, feats: torch.Tensor, batch_size: int, sizes=(25, 10), sampler=None, clip: float | None = None,
                 graph: bool = True, seed: int = 0, prefetch: bool = False):
        dev = feats.device
        self.model, self.opt, self.feats, self.sampler, self.clip = model, opt, feats, sampler, clip
        self.batch = DeviceBatch(batch_size, sizes if sampler is None else sampler.sizes, dev)
        self.seeds = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self.y = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self.state = StepState(dev, seed)
        self.params = [p for p in model.parameters()]
        self.loss = None
        self.logits = None
        self._one = torch.ones((), device=dev)
        self._side = torch.cuda.Stream(device=dev)                   # warm-up calls and the capture run here
        self._graph = None
        self._use_graph = graph
        self._lr = None
        self._calls = 0
        # epoch mode (set_epoch / step_epoch): the step draws its own seeds and labels from device-resident tables
        self._epoch_mode = False
        self._order = self._labels = None
        self._epoch_len = self._epoch_pos = 0
        # prefetch (sampler given, graph=False): the NEXT batch is sampled on a side stream while this step computes -- the
        # reference's DataLoader workers do the same on the host (main.py:100-104, persistent_workers).  Two batch buffers.
        self._prefetch = bool(prefetch and sampler is not None and not graph)
        if self._prefetch:
            self._batches = [self.batch, DeviceBatch(batch_size, sampler.sizes, dev)]
            self._ys = [self.y, torch.zeros_like(self.y)]
            self._sample_stream = torch.cuda.Stream(device=dev)
            self._ready = None          # (event, buffer index) of the batch sampled ahead
            self._free = [None, None]   # per buffer: event behind the last step that read it
        model.dropout_seed_dev = self.state.dropout_seed
        opt.use_device_step(self.state.adam_step)

    # ---- the step body: ordinary autograd code, capturable ----
    def _body(self, sample: bool = True):
        if self.sampler is not None and sample:
            if self._epoch_mode:
                self.sampler.sample_epoch_device(self._order, self.state.cursor, self._labels, self.y, seed=0, out=self.batch,
                                                 seed_dev=self.state.sample_seed)
            else:
                self.sampler.sample_device(self.seeds, seed=0, out=self.batch, seed_dev=self.state.sample_seed)
        x = IndexedFeatures(self.feats, self.batch.n_id)             # main.py:118-123 without the copy
        for p in self.params:
            p.grad = None
        logits = self.model(x, self.batch.adjs)
        loss = cross_entropy(logits, self.y, unit_upstream=True, loss_in=self.opt)   # main.py:216; the backward pass below is seeded with 1, the scalar is finished by opt.step()'s launch
        loss.backward(gradient=self._one)
        # detached views of the results: nothing outside this call keeps the autograd graph (and its AccumulateGrad nodes,
        # which remember the stream they were created on) alive into the next call or into the capture
        self.logits, self.loss = logits.detach(), loss.detach()
        del logits, loss
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip)   # main.py:286 gradient_clip_val
        self.opt.step()
        self.state.advance(sample_seed=sample, cursor_by=self.batch.n_seeds if self._epoch_mode else 0)

    def _capture(self):
        self.loss = self.logits = None
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self._side):
            self._body()
        self._graph = g
        self._lr = [grp["lr"] for grp in self.opt.param_groups]
        for tab in self.opt._tables.values():                        # the captured opt.step() counted a step that did not run
            tab["step"] -= 1
            for st in tab["states"]:
                st["step"] = tab["step"]

    def _run(self):
        self.model.train(True)
        if not self._use_graph:
            self._body()
            return
        if self._graph is not None and self._lr != [grp["lr"] for grp in self.opt.param_groups]:
            self._graph = None                                       # the scheduler changed lr: it is a launch argument
        if self._graph is None:
            if self._calls < 2:                                      # first calls eager: lazy initialisations (LDS opt-ins,
                self._calls += 1                                     # side streams, Adam state) happen outside the capture --
                cur = torch.cuda.current_stream()                    # on the stream the capture will use, as torch asks
                self._side.wait_stream(cur)
                with torch.cuda.stream(self._side), torch.autograd.set_multithreading_enabled(False):
                    self._body()
                cur.wait_stream(self._side)
                return
            with torch.autograd.set_multithreading_enabled(False):
                self._capture()
        self._graph.replay()
        # the host-side step counters of the optimiser follow the replays
        for tab in self.opt._tables.values():
            tab["step"] += 1
            for st in tab["states"]:
                st["step"] = tab["step"]

    def _sample_ahead(self, seeds, y, idx):
        """Enqueue the sampling of (seeds, y) into buffer idx on the side stream; returns the event behind it."""
        cur = torch.cuda.current_stream()
        side = self._sample_stream
        side.wait_stream(cur)                                        # seeds / y were produced on the caller's stream
        if self._free[idx] is not None:
            side.wait_event(self._free[idx])                         # the step that last read this buffer is done
        with torch.cuda.stream(side):
            copy_segments([self._ys[idx]], [y])
            self.sampler.sample_device(seeds.contiguous(), seed=0, out=self._batches[idx], seed_dev=self.state.sample_seed)
            self.state.advance_sample_seed()
            ev = torch.cuda.Event()
            ev.record(side)
        return ev

    def step(self, seeds: torch.Tensor, y: torch.Tensor, next_seeds: torch.Tensor | None = None, next_y: torch.Tensor | None = None):
        """Sample around `seeds` (device int64 [batch_size]) inside the step, labels `y` (device int64 [batch_size]).
        With prefetch=True, `next_seeds` / `next_y` (the batch of the NEXT call) are sampled on a side stream meanwhile."""
        assert self.sampler is not None, "no sampler: use load_batch() + run()"
        if self._epoch_mode:                                         # back to caller-provided seeds: another graph
            self._epoch_mode, self._graph, self._calls = False, None, 0
        if not self._prefetch:
            copy_segments([self.seeds, self.y], [seeds, y])
            self._run()
            return self.loss
        cur = torch.cuda.current_stream()
        if self._ready is None:                                      # first call: nothing was sampled ahead
            self._ready = (self._sample_ahead(seeds, y, 0), 0)
        ev, idx = self._ready
        cur.wait_event(ev)
        self.batch, self.y = self._batches[idx], self._ys[idx]
        self._ready = None
        if next_seeds is not None:
            self._ready = (self._sample_ahead(next_seeds, next_y, idx ^ 1), idx ^ 1)
        self.model.train(True)
        self._body(sample=False)
        done = torch.cuda.Event()
        done.record(cur)
        self._free[idx] = done
        return self.loss

    # ---- epoch mode: the loader's part of an epoch on the device too (main.py:100-123) ----
    def set_epoch(self, order: torch.Tensor, labels: torch.Tensor) -> None:
        """Start an epoch over `order` (device int64: the training nodes in this epoch's -- shuffled -- order; the reference's
        NeighborSampler(node_idx, batch_size, shuffle=True)) with `labels` (device int64 [N]: data.y).  :meth:`step_epoch` then
        trains on consecutive slices of batch_size nodes without any per-step input: the position lives in a device word the
        replayed step advances, the labels of a batch are gathered by the sampler's first kernel."""
        assert self.sampler is not None and order.is_cuda and order.dtype == torch.int64 and labels.is_cuda and labels.dtype == torch.int64
        if self._order is None or self._order.numel() < order.numel() or self._labels.numel() != labels.numel():
            self._order = torch.empty(max(order.numel(), 1), dtype=torch.int64, device=order.device)
            self._labels = torch.empty_like(labels)
            self._graph = None                                       # other buffers: the capture has to be redone
        if not self._epoch_mode:
            self._graph = None
            self._calls = 0
        self._epoch_mode = True
        copy_segments([self._order, self._labels, self.state.words[3:4]], [order.contiguous(), labels.contiguous(),
                                                                          torch.zeros(1, dtype=torch.int64, device=order.device)])
        self._epoch_len, self._epoch_pos = order.numel(), 0

    def batches_left(self) -> int:
        """Full batches the current epoch still holds (a shorter tail is the caller's, as in main.py's eager tail batch)."""
        return (self._epoch_len - self._epoch_pos) // self.batch.n_seeds

    def step_epoch(self):
        """One step on the next batch_size nodes of the epoch's order; the batch's node ids are self.batch.n_id[:batch_size]."""
        assert self._epoch_mode and self.batches_left() > 0, "set_epoch() first; the epoch's full batches are used up"
        self._run()
        self._epoch_pos += self.batch.n_seeds
        return self.loss

    def load_batch(self, pooled: DeviceBatch, y: torch.Tensor) -> None:
        """A batch of a pre-sampled pool (DeviceBatch.load) into the step's fixed buffers: one launch."""
        copy_segments(self.batch.segments() + [self.y], pooled.segments(valid_only=True) + [y])

    def run(self):
        """The step on whatever :meth:`load_batch` put into the buffers."""
        assert self.sampler is None
        self._run()
        return self.loss
