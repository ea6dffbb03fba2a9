"""Optimizers of the fine-tune step on MI355X (reference: ActiveLearning.py:219-231).

``AdamW`` has torch.optim.AdamW's constructor and param-group semantics (the reference builds
three groups with lr x{10, 1, 5}); ``step()`` is one ``vatl_adamw_step_multi`` launch per parameter group
(a device table of tensor pointers) instead of torch's element-wise kernels.  ``lr`` is read from the group at every step, so
``torch.optim.lr_scheduler.ExponentialLR`` works on it unchanged.
"""
from __future__ import annotations

import torch

import vatl_hip as vh


class AdamW(torch.optim.Optimizer):
    _kernel = staticmethod(vh.adamw_step)
    _multi = staticmethod(vh.adamw_step_multi)     # one launch per parameter group (161 tensors for SimplePose-R50)

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            batch = {}                                   # step count -> tensors: one multi-tensor launch per (group, step)
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                batch.setdefault(st["step"], []).append((p, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"]))
            for step, items in batch.items():
                if self._multi is not None and all(t[0].is_contiguous() for t in items):
                    self._multi([t[0].data for t in items], [t[1] for t in items], [t[2] for t in items], [t[3] for t in items], step,
                                group["lr"], group["weight_decay"], group["betas"], group["eps"])
                else:
                    for p, g, m, v in items:
                        self._kernel(p.data, g, m, v, step, group["lr"], group["weight_decay"], group["betas"], group["eps"])
                # the in-place update went through the C ABI: bump the version counters ourselves, the
                # inference plans key their packed-weight caches on them
                for p, *_ in items:
                    torch.autograd.graph.increment_version(p)
        return loss


class Adam(AdamW):
    """torch.optim.Adam (ActiveLearning.py:222-223): weight decay, if any, is an L2 term on the gradient."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)

    _kernel = staticmethod(vh.adam_step)
    _multi = None                                  # per-tensor launches (the reference's default optimiser is AdamW)


class SGD(torch.optim.Optimizer):
    """torch.optim.SGD with momentum as the reference configures it (ActiveLearning.py:220-221)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameters")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["momentum_buffer"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                vh.sgd_step(p.data, p.grad.contiguous(), st["momentum_buffer"], st["step"], group["lr"], group["momentum"],
                            group["weight_decay"])
                torch.autograd.graph.increment_version(p)
        return loss
