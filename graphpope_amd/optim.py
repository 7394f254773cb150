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
from ._lib import check


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._tables = {}            # per group: cached ctypes pointer arrays, rebuilt when the set of tensors changes

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
            step = steps.pop() + 1
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in live]
            key = tuple((p.data_ptr(), g.data_ptr()) for p, g in zip(live, grads))
            tab = self._tables.get(gi)
            if tab is None or tab[0] != key:          # gradients are fresh tensors every step: only their addresses change
                n = len(live)
                arr = ctypes.c_void_p * n
                tab = (key, n, arr(*[p.data_ptr() for p in live]), arr(*[g.data_ptr() for g in grads]),
                       arr(*[self.state[p]["exp_avg"].data_ptr() for p in live]),
                       arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in live]),
                       (ctypes.c_int64 * n)(*[p.numel() for p in live]))
                self._tables[gi] = tab
            _, n, pp, gg, mm, vv, nn = tab
            b1, b2 = group["betas"]
            dev = live[0].device
            with torch.cuda.device(dev):
                check(lib.sage_adam_step(n, pp, gg, mm, vv, nn, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                         float(group["weight_decay"]), step,
                                         ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
            for p in live:
                self.state[p]["step"] = step
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
