"""GraphSAGE on the HIP SAGEConv kernels: the consumer of the ``features (+) POPE`` tensor.

Mirrors /root/reference/main.py:182-211 (class SAGE: ModuleList of SAGEConv + BatchNorm1d, forward
over the sampled ``adjs``) with the PyG ``SAGEConv`` replaced by :class:`SAGEConv` below, whose
neighbour gather + mean and both projections run in libgraphpope_hip.so (fp32, exact-f32 MFMA).
Parameter names follow PyG 1.7.0 (``lin_l.weight``, ``lin_l.bias``, ``lin_r.weight``) so the
reference's Lightning checkpoints stay loadable.
"""
from __future__ import annotations

import ctypes
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from ._lib import check, on_device, ptr


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class SampledAdj:
    """One bipartite block of a sampled mini-batch: CSR by destination, destinations = first n_dst sources.

    Stands in for the ``torch_sparse.SparseTensor`` ``adj_t`` PyG's NeighborSampler yields
    (main.py:59-63, 118-123): ``size(0)`` = n_dst, ``size(1)`` = n_src, values dropped.

    ``dims`` (device int32 [4] = {n_dst, n_src, nnz, 0}, written by the device-extent sampler) makes the block one of
    DEVICE extents: ``n_dst`` / ``n_src`` / the lengths of ``rowptr`` and ``col`` are then capacities, the kernels read the
    true sizes from ``dims`` and nothing is ever read back to the host (include/graphpope_hip.h, "Device extents").
    """

    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, n_src: int, dims: torch.Tensor | None = None):
        self.rowptr = rowptr.to(torch.int32).contiguous()
        self.col = col.to(torch.int32).contiguous()
        self.n_dst = int(rowptr.numel() - 1)
        self.n_src = int(n_src)
        if dims is not None:
            assert dims.is_cuda and dims.dtype == torch.int32 and dims.numel() >= 3 and dims.is_contiguous()
        self.dims = dims

    def size(self, dim: int) -> int:
        return (self.n_dst, self.n_src)[dim]

    def to(self, device):
        return SampledAdj(self.rowptr.to(device), self.col.to(device), self.n_src, None if self.dims is None else self.dims.to(device))

    def device_extents(self) -> "SampledAdj":
        """The same block with its sizes in a device word triple (for blocks sampled on the host: pools of pre-sampled
        batches that feed a replayed step)."""
        if self.dims is not None:
            return self
        dims = torch.tensor([self.n_dst, self.n_src, int(self.col.numel()), 0], dtype=torch.int32, device=self.rowptr.device)
        return SampledAdj(self.rowptr, self.col, self.n_src, dims)


def _forward_scratch(lib, n_dst, c_in, c_out, dev):
    """Partial-tile slabs of the stream-K forward projection (None, 0 for layers too small to use it).  Taken from
    torch's caching allocator per call: stream-ordered like every other buffer of the step."""
    nbytes = lib.sage_conv_forward_scratch_bytes(n_dst, c_in, c_out)
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


def _colsum_of(grad_out):
    """The column sums of an incoming gradient if the op that produced it left them on the tensor (the fused BatchNorm backward,
    sage_bn_relu_dropout_backward_bias: ``_colsum``, float32 [C] for exactly this tensor), else None -- the layer's bias gradient
    without a pass over the gradient matrix."""
    cs = getattr(grad_out, "_colsum", None)
    if cs is not None and cs.is_cuda and cs.dtype == torch.float32 and cs.dim() == 1 and grad_out.dim() == 2 and cs.numel() == grad_out.shape[1]:
        return cs
    return None


def _stats_buffers(n_dst, c_out, dev):
    """Room for the per-row-tile column sums the projection's epilogue leaves for BatchNorm (sage_conv_forward_stats): float64
    [2, ceil(n_dst / 16), c_out] on the device and the two host ints the call reports (tiles written, rows per tile)."""
    return torch.empty((2, (n_dst + 15) // 16, c_out), dtype=torch.float64, device=dev), (ctypes.c_int32 * 2)(0, 0)


class _SageConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_src, w_l, b_l, w_r, rowptr, col, n_dst, dims=None, bn_stats=None):
        lib = _lib.load()
        if not x_src.is_cuda:
            raise RuntimeError("SAGEConv runs on the GPU only (no CPU fallback)")
        if not (x_src.is_contiguous() and w_l.is_contiguous() and w_r.is_contiguous()):
            x_src, w_l, w_r = x_src.contiguous(), w_l.contiguous(), w_r.contiguous()
        n_src, c_in = x_src.shape
        c_out = w_l.shape[0]
        agg = torch.empty((n_dst, c_in), dtype=torch.float32, device=x_src.device)
        out = torch.empty((n_dst, c_out), dtype=torch.float32, device=x_src.device)
        with on_device(x_src.device):
            scratch, nbytes = _forward_scratch(lib, n_dst, c_in, c_out, x_src.device)
            if bn_stats is not None:                  # the projection's epilogue leaves the first stage of the BatchNorm statistics of `out`
                stats, info = _stats_buffers(n_dst, c_out, x_src.device)
                check(lib.sage_conv_forward_stats(ptr(rowptr), ptr(col), n_src, n_dst, col.numel(), ptr(x_src), c_in, ptr(w_l),
                                                  ptr(b_l), ptr(w_r), c_out, ptr(agg), ptr(out), ptr(scratch), nbytes, ptr(dims),
                                                  ptr(stats[0]), ptr(stats[1]), stats.shape[1], info, _stream()))
                if info[0] > 0:
                    bn_stats.append((stats, int(info[0]), int(info[1])))
            else:
                check(lib.sage_conv_forward(ptr(rowptr), ptr(col), n_src, n_dst, col.numel(), ptr(x_src), c_in, ptr(w_l),
                                            ptr(b_l), ptr(w_r), c_out, ptr(agg), ptr(out), ptr(scratch), nbytes, ptr(dims), _stream()))
        ctx.save_for_backward(x_src, agg, w_l, w_r, rowptr, col)
        ctx.has_bias = b_l is not None
        ctx.n_dst = n_dst
        ctx.dims = dims
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x_src, agg, w_l, w_r, rowptr, col = ctx.saved_tensors
        n_src, c_in = x_src.shape
        c_out, n_dst = w_l.shape[0], ctx.n_dst
        given_b = _colsum_of(grad_out) if ctx.has_bias else None
        grad_out = grad_out.contiguous()
        dev = x_src.device
        need_x = ctx.needs_input_grad[0]
        grad_x = torch.empty_like(x_src) if need_x else None
        grad_w_l = torch.empty_like(w_l)
        grad_w_r = torch.empty_like(w_r)
        grad_b = torch.empty(c_out, dtype=torch.float32, device=dev) if (ctx.has_bias and given_b is None) else None
        with on_device(dev):
            scratch = torch.empty(max(lib.sage_conv_scratch_bytes(n_src, n_dst, col.numel(), c_in, c_out), 16),
                                  dtype=torch.uint8, device=dev)
            check(lib.sage_conv_backward(ptr(rowptr), ptr(col), n_src, n_dst, col.numel(), ptr(x_src), ptr(agg), c_in,
                                         ptr(w_l), ptr(w_r), c_out, ptr(grad_out), ptr(grad_x), ptr(grad_w_l), ptr(grad_b),
                                         ptr(grad_w_r), ptr(scratch), scratch.numel(), ptr(ctx.dims), _stream()))
        return grad_x, grad_w_l, (given_b if given_b is not None else grad_b), grad_w_r, None, None, None, None, None


class IndexedFeatures:
    """``Batch.x = data.x[n_id]`` (main.py:118-123) WITHOUT the copy: the resident feature matrix plus the batch's node
    ids.  Passed to :class:`SAGE` / :class:`SAGEConv` in place of the gathered ``x``, the first layer reads its
    neighbours' rows straight from ``feats`` (sage_conv_forward_indexed); only the destination rows are materialised."""

    def __init__(self, feats: torch.Tensor, n_id: torch.Tensor):
        assert feats.dim() == 2 and feats.dtype == torch.float32 and feats.is_contiguous() and not feats.requires_grad
        assert n_id.dtype == torch.int64 and n_id.dim() == 1 and n_id.device == feats.device
        self.feats, self.n_id = feats, n_id.contiguous()

    def materialize(self) -> torch.Tensor:
        return self.feats.index_select(0, self.n_id)


class _SageConvIndexedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w_l, b_l, w_r, feats, n_id, rowptr, col, n_dst, dims=None, bn_stats=None):
        lib = _lib.load()
        if not feats.is_cuda:
            raise RuntimeError("SAGEConv runs on the GPU only (no CPU fallback)")
        if not (w_l.is_contiguous() and w_r.is_contiguous()):
            w_l, w_r = w_l.contiguous(), w_r.contiguous()
        c_in, c_out, dev = feats.shape[1], w_l.shape[0], feats.device
        agg = torch.empty((n_dst, c_in), dtype=torch.float32, device=dev)
        x_dst = torch.empty((n_dst, c_in), dtype=torch.float32, device=dev)
        out = torch.empty((n_dst, c_out), dtype=torch.float32, device=dev)
        with on_device(dev):
            scratch, nbytes = _forward_scratch(lib, n_dst, c_in, c_out, dev)
            if bn_stats is not None:
                stats, info = _stats_buffers(n_dst, c_out, dev)
                check(lib.sage_conv_forward_indexed_stats(ptr(rowptr), ptr(col), ptr(n_id), n_id.numel(), n_dst, col.numel(), ptr(feats),
                                                          feats.shape[0], c_in, ptr(w_l), ptr(b_l), ptr(w_r), c_out, ptr(agg), ptr(x_dst),
                                                          ptr(out), ptr(scratch), nbytes, ptr(dims), ptr(stats[0]), ptr(stats[1]),
                                                          stats.shape[1], info, _stream()))
                if info[0] > 0:
                    bn_stats.append((stats, int(info[0]), int(info[1])))
            else:
                check(lib.sage_conv_forward_indexed(ptr(rowptr), ptr(col), ptr(n_id), n_id.numel(), n_dst, col.numel(), ptr(feats),
                                                    feats.shape[0], c_in, ptr(w_l), ptr(b_l), ptr(w_r), c_out, ptr(agg), ptr(x_dst),
                                                    ptr(out), ptr(scratch), nbytes, ptr(dims), _stream()))
        # x_dst is kept as a matrix for the backward pass: reading the destination rows through n_id in the weight-gradient kernel
        # (sage_conv_backward_indexed) was measured 29 us slower per step than the 60 MB this copy costs
        ctx.save_for_backward(x_dst, agg, w_l, w_r, rowptr, col)
        ctx.has_bias = b_l is not None
        ctx.dims = dims
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x_dst, agg, w_l, w_r, rowptr, col = ctx.saved_tensors
        n_dst, c_in = x_dst.shape
        c_out, dev = w_l.shape[0], x_dst.device
        given_b = _colsum_of(grad_out) if ctx.has_bias else None
        grad_out = grad_out.contiguous()
        grad_w_l, grad_w_r = torch.empty_like(w_l), torch.empty_like(w_r)
        grad_b = torch.empty(c_out, dtype=torch.float32, device=dev) if (ctx.has_bias and given_b is None) else None
        with on_device(dev):
            scratch = torch.empty(max(lib.sage_conv_scratch_bytes(n_dst, n_dst, col.numel(), c_in, c_out), 16), dtype=torch.uint8, device=dev)
            check(lib.sage_conv_backward(ptr(rowptr), ptr(col), n_dst, n_dst, col.numel(), ptr(x_dst), ptr(agg), c_in, ptr(w_l),
                                         ptr(w_r), c_out, ptr(grad_out), None, ptr(grad_w_l), ptr(grad_b), ptr(grad_w_r),
                                         ptr(scratch), scratch.numel(), ptr(ctx.dims), _stream()))
        return grad_w_l, (given_b if given_b is not None else grad_b), grad_w_r, None, None, None, None, None, None, None


class _Linear(nn.Module):
    """Weight (+ bias) holder named like torch.nn.Linear so state dicts line up with PyG's lin_l / lin_r."""

    def __init__(self, c_in, c_out, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c_out, c_in))
        self.bias = nn.Parameter(torch.empty(c_out)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))       # torch.nn.Linear's default
        if self.bias is not None:
            bound = 1 / math.sqrt(self.weight.shape[1])
            nn.init.uniform_(self.bias, -bound, bound)


class SAGEConv(nn.Module):
    """``conv((x_src, x_dst), adj_t)`` with mean aggregation (PyG 1.7.0 SAGEConv defaults: root_weight, bias, no normalize)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin_l = _Linear(in_channels, out_channels, bias=True)
        self.lin_r = _Linear(in_channels, out_channels, bias=False)

    def forward(self, x, adj_t: SampledAdj, bn_stats: bool = False):
        """``bn_stats``: the output goes straight into :func:`bn_relu_dropout` in training mode -- the projection's epilogue then also
        produces the first stage of its BatchNorm statistics (carried on the returned tensor as ``_bn_stats``), one launch less."""
        holder = [] if bn_stats else None                         # the autograd function appends (stats, tiles, rows per tile) when its kernels produced them
        if isinstance(x, IndexedFeatures):                        # neighbours read straight from the resident feature matrix
            out = _SageConvIndexedFn.apply(self.lin_l.weight, self.lin_l.bias, self.lin_r.weight, x.feats, x.n_id, adj_t.rowptr,
                                           adj_t.col, adj_t.size(0), adj_t.dims, holder)
        else:
            x_src = x[0] if isinstance(x, (tuple, list)) else x   # x_dst = x_src[:n_dst] by construction (main.py:206)
            out = _SageConvFn.apply(x_src, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight, adj_t.rowptr, adj_t.col,
                                    adj_t.size(0), adj_t.dims, holder)
        if holder:
            out._bn_stats = holder[0]
        return out


class _BnReluDropoutFn(torch.autograd.Function):
    """BatchNorm1d -> ReLU -> dropout in three launches per direction (csrc/epilogue.hip), main.py:207-209."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, training, p, seed, num_batches_tracked=None,
                rows=None, seed_dev=None, stats=None):
        lib = _lib.load()
        if not x.is_cuda:
            raise RuntimeError("the fused BatchNorm/ReLU/dropout epilogue runs on the GPU only (no CPU fallback)")
        x = x.contiguous()
        m, c = x.shape
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        rstd = torch.empty(c, dtype=torch.float32, device=dev)
        with on_device(dev):
            scratch = torch.empty(lib.sage_bn_scratch_bytes(c), dtype=torch.uint8, device=dev)
            if stats is not None and training:        # the first stage of the statistics came with x (sage_conv_forward_stats)
                st, parts, rows_per_part = stats
                check(lib.sage_bn_relu_dropout_forward_stats(ptr(x), m, c, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                                             ptr(num_batches_tracked), momentum, eps, int(training), p, seed, ptr(y), ptr(mean),
                                                             ptr(rstd), ptr(scratch), scratch.numel(), ptr(rows), ptr(seed_dev), ptr(st[0]),
                                                             ptr(st[1]), parts, rows_per_part, _stream()))
            else:
                check(lib.sage_bn_relu_dropout_forward(ptr(x), m, c, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                                       ptr(num_batches_tracked), momentum, eps, int(training), p, seed, ptr(y), ptr(mean), ptr(rstd),
                                                       ptr(scratch), scratch.numel(), ptr(rows), ptr(seed_dev), _stream()))
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (bool(training), float(p), int(seed))
        ctx.dev_words = (rows, seed_dev)
        ctx.producer_bias = stats is not None         # x came from a SAGEConv that cooperates: hand it its bias gradient too (backward)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        lib = _lib.load()
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        training, p, seed = ctx.cfg
        m, c = x.shape
        dev = x.device
        grad_y = grad_y.contiguous()
        grad_x = torch.empty_like(x)
        grad_gamma = torch.empty_like(gamma)
        grad_beta = torch.empty_like(beta)
        with on_device(dev):
            scratch = torch.empty(lib.sage_bn_scratch_bytes(c), dtype=torch.uint8, device=dev)
            if ctx.producer_bias:
                # the column sums of grad_x out of the statistics pass's float64 sums: the producing layer's bias gradient, which
                # its own backward pass would otherwise get by reading grad_x back (one launch); it finds them on the tensor
                colsum = torch.empty(c, dtype=torch.float32, device=dev)
                check(lib.sage_bn_relu_dropout_backward_bias(ptr(x), ptr(grad_y), m, c, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                                             int(training), p, seed, ptr(grad_x), ptr(grad_gamma), ptr(grad_beta),
                                                             ptr(scratch), scratch.numel(), ptr(ctx.dev_words[0]), ptr(ctx.dev_words[1]),
                                                             ptr(colsum), _stream()))
                grad_x._colsum = colsum
            else:
                check(lib.sage_bn_relu_dropout_backward(ptr(x), ptr(grad_y), m, c, ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                                        int(training), p, seed, ptr(grad_x), ptr(grad_gamma), ptr(grad_beta),
                                                        ptr(scratch), scratch.numel(), ptr(ctx.dev_words[0]), ptr(ctx.dev_words[1]), _stream()))
        return grad_x, grad_gamma, grad_beta, None, None, None, None, None, None, None, None, None, None, None


def bn_relu_dropout(x: torch.Tensor, bn: nn.BatchNorm1d, p: float, training: bool, seed: int | None = None,
                    rows: torch.Tensor | None = None, seed_dev: torch.Tensor | None = None) -> torch.Tensor:
    """``F.dropout(bn(x).relu_(), p, training)`` (main.py:207-209) on the fused HIP epilogue.

    `bn` stays an ordinary ``nn.BatchNorm1d`` (same state-dict keys as the reference's checkpoints); its running
    statistics and ``num_batches_tracked`` are updated as torch does.  The dropout mask is a counter hash of
    (seed, element): `seed` defaults to a draw from torch's global generator, so ``torch.manual_seed`` makes runs
    repeatable; the mask itself is not torch's Philox stream.

    ``rows`` (device int32 scalar): the true row count of ``x`` when its first dimension is a capacity (device-extent
    batches).  ``seed_dev`` (device int64 scalar): added to ``seed`` on the device, so that a replayed HIP graph draws a
    new mask every replay; with it ``seed`` defaults to 0 instead of a draw from torch's generator (which is a host read).
    """
    if bn.weight is None or bn.momentum is None:
        raise NotImplementedError("fused epilogue: affine BatchNorm1d with a fixed momentum only (the reference's default)")
    use_batch_stats = training or bn.running_mean is None
    nbt = bn.num_batches_tracked if (training and bn.track_running_stats and bn.num_batches_tracked is not None) else None
    if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64):
        nbt.add_(1)                                       # a counter the kernel cannot reach: torch's own op
        nbt = None
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (training and p > 0 and seed_dev is None) else 0
    rm = bn.running_mean if bn.track_running_stats else None
    rv = bn.running_var if bn.track_running_stats else None
    stats = getattr(x, "_bn_stats", None) if (use_batch_stats and x.is_contiguous()) else None      # left by SAGEConv(..., bn_stats=True) for exactly this tensor
    out = _BnReluDropoutFn.apply(x, bn.weight, bn.bias, rm, rv, float(bn.momentum), float(bn.eps), use_batch_stats,
                                 float(p) if training else 0.0, seed, nbt, rows, seed_dev, stats)       # the step counter goes up inside the statistics kernel
    return out


_BAD_LABEL = {}          # per device: int32 [1], set to 1 by the kernel when a label is out of range (never cleared here)


def bad_label_flag(device) -> torch.Tensor:
    """Device int32 [1]: non-zero once :func:`cross_entropy` has met a label outside [0, C) that is not ignore_index
    (such rows are left out of the mean, as ignored ones; torch raises a device-side assert instead)."""
    dev = torch.device(device)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    if key not in _BAD_LABEL:
        _BAD_LABEL[key] = torch.zeros(1, dtype=torch.int32, device=dev)
    return _BAD_LABEL[key]


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index, unit_upstream=False, loss_in=None):
        lib = _lib.load()
        if not logits.is_cuda:
            raise RuntimeError("cross_entropy runs on the GPU only (no CPU fallback)")
        logits = logits.contiguous()
        n, c = logits.shape
        dev = logits.device
        out = torch.empty(2, dtype=torch.float32, device=dev)                # [loss, 1 / count]
        grad = torch.empty_like(logits)
        rows = torch.empty(n, dtype=torch.float32, device=dev)
        with on_device(dev):
            check(lib.sage_cross_entropy_forward(ptr(logits), ptr(target), n, c, ignore_index, ptr(out), ptr(grad),
                                                 ctypes.c_void_p(out.data_ptr() + 4), ptr(rows), ptr(bad_label_flag(dev)),
                                                 (2 if loss_in is not None else 1) if unit_upstream else 0, _stream()))
        if unit_upstream and loss_in is not None:
            loss_in.fold_loss(rows, out)              # the optimiser's launch finishes the scalar (sage_adam_step_loss)
        ctx.save_for_backward(grad, out)
        ctx.unit_upstream = bool(unit_upstream)
        return out[0]

    @staticmethod
    def backward(ctx, grad_loss):
        lib = _lib.load()
        grad, out = ctx.saved_tensors
        if ctx.unit_upstream:                 # the forward launch already produced d(mean loss) / d(logits): nothing to launch
            return grad, None, None, None, None
        n, c = grad.shape
        grad_loss = grad_loss.contiguous()
        res = torch.empty_like(grad)
        with on_device(grad.device):
            check(lib.sage_cross_entropy_backward(ptr(grad), n, c, ptr(grad_loss), ctypes.c_void_p(out.data_ptr() + 4), ptr(res),
                                                  _stream()))
        return res, None, None, None, None


def cross_entropy(logits: torch.Tensor, target: torch.Tensor, ignore_index: int = -100, unit_upstream: bool = False, loss_in=None) -> torch.Tensor:
    """``F.cross_entropy(logits, target)`` (main.py:216: mean over the rows, integer labels) in two launches forward and
    one backward: the softmax - onehot gradient is produced with the loss and only scaled in the backward pass.

    ``unit_upstream=True`` is a promise that the loss is the root of the backward pass and is seeded with a gradient of 1
    (``loss.backward()``): the whole forward pass is then ONE launch that also scales the gradient by 1 / count, and the
    backward pass launches nothing.  Any other upstream gradient would be ignored -- only a training step that owns its
    ``backward()`` call (graphpope_amd.train.SageTrainStep) sets it.

    ``loss_in`` (a graphpope_amd.optim.Adam, with ``unit_upstream``): the last stage of the forward pass -- the mean of the row
    losses -- becomes one more block of that optimiser's next ``step()`` launch instead of a launch of its own in front of the
    backward pass; the returned scalar is valid once that step has run (round 5: one launch and one kernel boundary less per step)."""
    if target.dtype != torch.int64 or target.dim() != 1 or logits.dim() != 2 or target.shape[0] != logits.shape[0]:
        raise ValueError("cross_entropy: logits [N, C] float32 and int64 labels [N] expected")
    if loss_in is not None and not unit_upstream:
        raise ValueError("cross_entropy: loss_in needs unit_upstream=True (a step that owns its backward() and step() calls)")
    return _CrossEntropyFn.apply(logits, target.contiguous(), int(ignore_index), bool(unit_upstream), loss_in)


class SAGE(nn.Module):
    """main.py:182-211 without the Lightning plumbing.  Keeps the reference's depth quirk: ``forward`` iterates over
    the sampled adjs (two of them, sizes=[25, 10]), so with num_layers=3 the last conv / bn are never executed and
    the logits are hidden_channels wide (SURVEY.md §7 trap 7)."""

    def __init__(self, in_channels: int, out_channels: int, hidden_channels: int, num_layers: int, dropout: float = 0.5):
        super().__init__()
        self.dropout = dropout
        self.convs = nn.ModuleList()
        self.convs.append(SAGEConv(in_channels, hidden_channels))
        for _ in range(num_layers - 2):
            self.convs.append(SAGEConv(hidden_channels, hidden_channels))
        self.convs.append(SAGEConv(hidden_channels, out_channels))
        self.bns = nn.ModuleList()
        for _ in range(num_layers - 1):
            self.bns.append(nn.BatchNorm1d(hidden_channels))
        self.dropout_seed_dev = None     # device int64 scalar the dropout seeds follow (train.SageTrainStep sets it: graph replay)

    def forward(self, x, adjs):
        for i, adj_t in enumerate(adjs):
            to_bn = i < len(adjs) - 1
            x = self.convs[i](x if isinstance(x, IndexedFeatures) else (x, x[:adj_t.size(0)]), adj_t, bn_stats=to_bn and self.training)
            if to_bn:
                rows = None if adj_t.dims is None else adj_t.dims[0:1]
                if self.dropout_seed_dev is not None:
                    x = bn_relu_dropout(x, self.bns[i], self.dropout, self.training, seed=0x9E3779B97F4A7C15 * (i + 1) % (1 << 63),
                                        rows=rows, seed_dev=self.dropout_seed_dev)
                else:
                    x = bn_relu_dropout(x, self.bns[i], self.dropout, self.training, rows=rows)
        return x


# ------------------------------------------------------------------------------------------------
# Host-side fan-out sampler for synthetic batches (stands in for PyG NeighborSampler, main.py:100-116).
# Not accelerated; only used to produce Flickr-shaped pre-sampled batches.
# ------------------------------------------------------------------------------------------------
def sample_batch(rowptr: np.ndarray, col: np.ndarray, seeds: np.ndarray, sizes=(25, 10), rng=None):
    """Returns (n_id, adjs) like NeighborSampler: adjs outer -> inner, each a SampledAdj over local ids, and the
    destinations of every block are the first n_dst entries of its sources."""
    rng = rng or np.random.default_rng(0)
    n_id = np.asarray(seeds, dtype=np.int64)
    adjs = []
    for size in sizes:
        local = {int(g): i for i, g in enumerate(n_id)}
        ids = list(n_id)
        rp = [0]
        cols = []
        for g in n_id:
            nbr = col[rowptr[g]:rowptr[g + 1]]
            if nbr.size > size:
                nbr = rng.choice(nbr, size, replace=False)
            for u in nbr:
                u = int(u)
                j = local.get(u)
                if j is None:
                    j = len(ids)
                    local[u] = j
                    ids.append(u)
                cols.append(j)
            rp.append(len(cols))
        adjs.append(SampledAdj(torch.tensor(rp, dtype=torch.int32), torch.tensor(cols, dtype=torch.int32), len(ids)))
        n_id = np.asarray(ids, dtype=np.int64)
    return n_id, adjs[::-1]
