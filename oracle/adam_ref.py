"""CPU oracle for the optimiser step.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

PINNED: the arithmetic is ``torch.optim.Adam`` itself -- the very implementation the reference calls
(``nerfstudio/engine/optimizers.py:57-63``: ``AdamOptimizerConfig._target = torch.optim.Adam``) and a package
of this image -- driven the way the reference drives it: one optimiser per named group
(``optimizers.py:82-115``), ``LambdaLR`` with the exponential-decay multiplier (``schedulers.py:116-141``,
restated below), gradients zeroed and groups stepped by their accumulation periods (``trainer.py:494-510``).
"""
from __future__ import annotations

import numpy as np
import torch


def exponential_decay_lambda(lr_init, lr_final=None, max_steps=100000, warmup_steps=0, lr_pre_warmup=1e-8,
                             ramp="cosine"):
    """schedulers.py:116-139."""
    if lr_final is None:
        lr_final = lr_init

    def func(step):
        if step < warmup_steps:
            if ramp == "cosine":
                lr = lr_pre_warmup + (lr_init - lr_pre_warmup) * np.sin(0.5 * np.pi * np.clip(step / warmup_steps, 0, 1))
            else:
                lr = lr_pre_warmup + (lr_init - lr_pre_warmup) * step / warmup_steps
        else:
            t = np.clip((step - warmup_steps) / (max_steps - warmup_steps), 0, 1)
            lr = np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
        return lr / lr_init

    return func


class Optimizers:
    """config: {group: {"lr", "eps", "scheduler": dict(lr_final=, max_steps=, ...) or None}}."""

    def __init__(self, config, param_groups):
        self.optimizers, self.schedulers, self.parameters = {}, {}, {}
        for name, params in param_groups.items():
            cfg = config[name]
            self.optimizers[name] = torch.optim.Adam(params, lr=cfg["lr"], eps=cfg.get("eps", 1e-15))
            self.parameters[name] = params
            if cfg.get("scheduler"):
                self.schedulers[name] = torch.optim.lr_scheduler.LambdaLR(
                    self.optimizers[name], lr_lambda=exponential_decay_lambda(cfg["lr"], **cfg["scheduler"]))

    def train_iteration(self, step, accumulation, set_grads):
        """trainer.py:494-510 with the loss replaced by ``set_grads(name, param_index) -> gradient``, which is
        ACCUMULATED into ``.grad`` as ``backward()`` would."""
        for name in self.parameters:
            if step % accumulation.get(name, 1) == 0:
                self.optimizers[name].zero_grad()
        for name, params in self.parameters.items():
            for i, p in enumerate(params):
                g = set_grads(name, i)
                if g is not None:
                    p.grad = g.clone() if p.grad is None else p.grad + g
        for name in self.parameters:
            a = accumulation.get(name, 1)
            if step % a == a - 1 and any(p.grad is not None for p in self.parameters[name]):
                self.optimizers[name].step()
        for sched in self.schedulers.values():
            sched.step()
