"""torch.optim.Adam with the whole step in ONE kernel launch (csrc/epilogue.hip: sage_adam_step).

The reference's optimiser is ``torch.optim.Adam(self.parameters(), lr=args.lr)`` (/root/reference/main.py:244).
torch's default implementation costs eight foreach launches and ~80 us of Python per step; GraphSAGE's training step on
MI355X is launch-bound, so this class keeps torch's interface (``param_groups`` for ``ReduceLROnPlateau``,
``state_dict`` keys ``step`` / ``exp_avg`` / ``exp_avg_sq``) and replaces the arithmetic by one call into the library.
Same update rule (amsgrad off, maximize off); float32 CUDA parameters only -- it raises otherwise, there is no fallback.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import check, on_device


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._tables = {}            # per group: cached ctypes pointer arrays, rebuilt when the set of tensors changes
        self.step_dev = None         # device int64 scalar holding the 1-based step count (use_device_step)
        self._pending_loss = None    # (row losses, [loss, 1 / count]) of a cross-entropy whose last stage rides in the next step's launch

    def use_device_step(self, word: torch.Tensor | None) -> None:
        """Read the step count of the bias corrections from a device word instead of the host counter: a training step that
        is replayed as a HIP graph cannot take it as a launch argument.  The word must hold the number of the step being
        taken (the caller advances it, sage_advance_counters); the host-side counters keep counting the calls made here."""
        assert word is None or (word.is_cuda and word.dtype == torch.int64 and word.numel() == 1)
        self.step_dev = word

    def fold_loss(self, row_losses: torch.Tensor, out: torch.Tensor) -> None:
        """The next :meth:`step` finishes a cross-entropy in its own launch (sage_adam_step_loss): ``row_losses`` (float32 [N], one
        loss per row, written by sage_cross_entropy_forward with fused = 2) -> ``out`` (float32 [2]: mean loss, 1 / count).  The
        scalar is only valid once that step has run -- graphpope_amd.sage.cross_entropy(loss_in=optimizer) arranges this for a
        training step that owns its backward() and step() calls."""
        assert row_losses.is_cuda and row_losses.dtype == torch.float32 and row_losses.is_contiguous() and out.numel() >= 2
        self._pending_loss = (row_losses, out)

    def _build(self, group, live):
        """Slow path, taken when the set of parameters with gradients changes: checks, state creation, pointer tables."""
        for p in live:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise RuntimeError("graphpope_amd.optim.Adam: float32 contiguous CUDA parameters only (no CPU fallback)")
            st = self.state[p]
            if not st:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        steps = {int(self.state[p]["step"]) for p in live}
        if len(steps) != 1:
            raise RuntimeError("graphpope_amd.optim.Adam: the parameters of a group must share one step count")
        n = len(live)
        arr = ctypes.c_void_p * n
        return {"live": live, "ids": [id(p) for p in live], "pptr": [p.data_ptr() for p in live], "n": n, "step": steps.pop(),
                "pp": arr(*[p.data_ptr() for p in live]), "gg": arr(*([0] * n)), "gptr": [0] * n,
                "mm": arr(*[self.state[p]["exp_avg"].data_ptr() for p in live]),
                "vv": arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in live]),
                "nn": (ctypes.c_int64 * n)(*[p.numel() for p in live]), "dev": live[0].device,
                "states": [self.state[p] for p in live]}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            tab = self._tables.get(gi)
            if (tab is None or tab["n"] != len(live) or any(a != id(b) for a, b in zip(tab["ids"], live))
                    or any(a != b.data_ptr() for a, b in zip(tab["pptr"], live))):      # e.g. module.to() swapped the storage
                tab = self._tables[gi] = self._build(group, live)
            keep = []                                         # gradients are fresh tensors every step: refresh their addresses
            gptr, gg = tab["gptr"], tab["gg"]
            for i, p in enumerate(live):
                g = p.grad
                if not g.is_contiguous():
                    g = g.contiguous()
                    keep.append(g)
                a = g.data_ptr()
                if a != gptr[i]:
                    gptr[i] = a
                    gg[i] = a
            step = tab["step"] = tab["step"] + 1
            b1, b2 = group["betas"]
            pending, self._pending_loss = self._pending_loss, None
            with on_device(tab["dev"]):
                if pending is None:
                    check(lib.sage_adam_step(tab["n"], tab["pp"], gg, tab["mm"], tab["vv"], tab["nn"], float(group["lr"]), float(b1),
                                             float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                             ctypes.c_void_p(0 if self.step_dev is None else self.step_dev.data_ptr()),
                                             ctypes.c_void_p(torch.cuda.current_stream(tab["dev"]).cuda_stream)))
                else:                                         # this launch also finishes the step's cross-entropy (one block more)
                    check(lib.sage_adam_step_loss(tab["n"], tab["pp"], gg, tab["mm"], tab["vv"], tab["nn"], float(group["lr"]), float(b1),
                                                  float(b2), float(group["eps"]), float(group["weight_decay"]), step,
                                                  ctypes.c_void_p(0 if self.step_dev is None else self.step_dev.data_ptr()),
                                                  ctypes.c_void_p(pending[0].data_ptr()), pending[0].numel(), ctypes.c_void_p(pending[1].data_ptr()),
                                                  ctypes.c_void_p(torch.cuda.current_stream(tab["dev"]).cuda_stream)))
            for st in tab["states"]:
                st["step"] = step
        if self._pending_loss is not None:                   # no parameter had a gradient: finish the loss on its own
            pending, self._pending_loss = self._pending_loss, None
            with on_device(pending[0].device):
                check(lib.sage_adam_step_loss(0, None, None, None, None, None, 0.0, 0.0, 0.0, 0.0, 0.0, 1, None,
                                              ctypes.c_void_p(pending[0].data_ptr()), pending[0].numel(), ctypes.c_void_p(pending[1].data_ptr()),
                                              ctypes.c_void_p(torch.cuda.current_stream(pending[0].device).cuda_stream)))
        return loss

    def state_dict(self):
        """torch.optim.Adam's layout: ``step`` is a float32 CPU tensor per parameter."""
        sd = super().state_dict()
        sd["state"] = {k: {**st, "step": torch.tensor(float(st["step"]))} if "step" in st else dict(st)
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = int(st["step"].item())
        self._tables.clear()
