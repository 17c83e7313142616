"""Fused Adam over parameter groups on MI355X: the optimiser step of the POGS trainer.

Mirrors nerfstudio's ``Optimizers`` as the trainer drives it
(``nerfstudio/engine/optimizers.py:74-193``, ``nerfstudio/engine/trainer.py:494-510``) for the configuration
POGS uses (``pogs/pogs_config.py:27,38-85``): one ``torch.optim.Adam(lr, eps=1e-15)`` per named group, an
optional exponential-decay schedule per group (``nerfstudio/engine/schedulers.py:92-141``) and per-group
gradient-accumulation periods.  Same method names and meaning; the difference is that all groups due in an
iteration are updated by ONE kernel launch (``pr_adam_step``).  No PyTorch fallback.

State is kept in torch's own layout, so ``state_dict()[name]`` loads into ``torch.optim.Adam`` and back.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, List, Optional

import torch
from torch import Tensor

from . import _lib
from .rendering import _call, _stream


class ExponentialDecay:
    """Learning-rate multiplier of ``ExponentialDecayScheduler`` (``schedulers.py:116-141``): optional warm-up
    from ``lr_pre_warmup``, then log-linear interpolation from ``lr_init`` to ``lr_final`` over ``max_steps``."""

    def __init__(self, lr_final: Optional[float] = None, max_steps: int = 100000, warmup_steps: int = 0,
                 lr_pre_warmup: float = 1e-8, ramp: str = "cosine"):
        self.lr_final, self.max_steps, self.warmup_steps = lr_final, max_steps, warmup_steps
        self.lr_pre_warmup, self.ramp = lr_pre_warmup, ramp

    def lr(self, lr_init: float, step: int) -> float:
        lr_final = lr_init if self.lr_final is None else self.lr_final
        if step < self.warmup_steps:
            if self.ramp == "cosine":
                return self.lr_pre_warmup + (lr_init - self.lr_pre_warmup) * math.sin(
                    0.5 * math.pi * min(max(step / self.warmup_steps, 0.0), 1.0))
            return self.lr_pre_warmup + (lr_init - self.lr_pre_warmup) * step / self.warmup_steps
        t = min(max((step - self.warmup_steps) / (self.max_steps - self.warmup_steps), 0.0), 1.0)
        return math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)


class Optimizers:
    """config: ``{group: {"lr": float, "eps": float = 1e-15, "betas": (0.9, 0.999), "scheduler": ExponentialDecay
    or None}}``; param_groups: ``{group: [parameters]}`` (float32 HIP tensors)."""

    def __init__(self, config: Dict[str, dict], param_groups: Dict[str, List[Tensor]]):
        _lib.load()
        self.config, self.parameters = config, {}
        self._state: Dict[str, List[dict]] = {}
        self._sched_steps: Dict[str, int] = {}
        for name, params in param_groups.items():
            if name not in config:
                raise RuntimeError(f"Optimizer config for '{name}' not found in config file. Provided configs "
                                   f"were: {config.keys()}")
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("pogs_amd.optim needs contiguous float32 parameters on a HIP device "
                                       "(no CPU fallback)")
            self.parameters[name] = list(params)
            self._state[name] = [{} for _ in params]
            self._sched_steps[name] = 0

    # ---- learning rates ------------------------------------------------------------------------------------
    def lr(self, name: str) -> float:
        cfg = self.config[name]
        sched = cfg.get("scheduler")
        return cfg["lr"] if sched is None else sched.lr(cfg["lr"], self._sched_steps[name])

    def scheduler_step(self, name: str) -> None:
        if self.config[name].get("scheduler") is not None:
            self._sched_steps[name] += 1

    def scheduler_step_all(self, step: int) -> None:
        for name in self.parameters:
            self.scheduler_step(name)

    # ---- gradients -----------------------------------------------------------------------------------------
    def zero_grad_all(self) -> None:
        self.zero_grad_some(list(self.parameters))

    def zero_grad_some(self, param_groups: List[str]) -> None:
        for name in param_groups:
            for p in self.parameters[name]:
                p.grad = None

    # ---- steps ---------------------------------------------------------------------------------------------
    def optimizer_step(self, param_group_name: str) -> None:
        self.optimizer_step_some([param_group_name])

    def optimizer_step_all(self) -> None:
        self.optimizer_step_some(list(self.parameters))

    def optimizer_step_some(self, param_groups: List[str]) -> None:
        """One Adam update of every parameter (that has a gradient) of the named groups, in one launch per
        ``pr_adam_max_groups()`` tensors."""
        entries, keep = [], []
        for name in param_groups:
            cfg = self.config[name]
            b1, b2 = cfg.get("betas", (0.9, 0.999))
            lr, eps = self.lr(name), cfg.get("eps", 1e-15)
            for p, st in zip(self.parameters[name], self._state[name]):
                if p.grad is None:
                    continue
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() % 16:
                    g = g.float().contiguous().clone()
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                keep.append(g)
                entries.append(_lib.AdamGroup(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                              st["exp_avg_sq"].data_ptr(), p.numel(), st["step"], lr, b1, b2, eps))
        if not entries:
            return
        lib = _lib.load()
        chunk = lib.pr_adam_max_groups()
        dev = keep[0].device
        with torch.cuda.device(dev), torch.no_grad():
            for i in range(0, len(entries), chunk):
                part = entries[i:i + chunk]
                arr = (_lib.AdamGroup * len(part))(*part)
                _call("pr_adam_step", ctypes.cast(arr, ctypes.c_void_p), len(part), _stream())
        del keep

    # the trainer's per-iteration bookkeeping (trainer.py:494-510)
    @staticmethod
    def groups_to_zero(step: int, names, accumulation: Dict[str, int]) -> List[str]:
        return [n for n in names if step % accumulation.get(n, 1) == 0]

    @staticmethod
    def groups_to_step(step: int, names, accumulation: Dict[str, int]) -> List[str]:
        return [n for n in names if step % accumulation.get(n, 1) == accumulation.get(n, 1) - 1]

    # ---- checkpoints: torch.optim.Adam's own layout ----------------------------------------------------------
    def state_dict(self) -> Dict[str, dict]:
        out = {}
        for name, params in self.parameters.items():
            cfg = self.config[name]
            state = {i: {"step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"],
                         "exp_avg_sq": st["exp_avg_sq"]}
                     for i, st in enumerate(self._state[name]) if st}
            group = {"lr": self.lr(name), "betas": tuple(cfg.get("betas", (0.9, 0.999))), "eps": cfg.get("eps", 1e-15),
                     "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                     "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                     "initial_lr": cfg["lr"], "params": list(range(len(params)))}
            out[name] = {"state": state, "param_groups": [group]}
        return out

    def load_optimizers(self, loaded_state: Dict[str, dict]) -> None:
        for name, sd in loaded_state.items():
            for i, st in sd["state"].items():
                p = self.parameters[name][int(i)]
                self._state[name][int(i)] = {
                    "step": int(float(st["step"])),
                    "exp_avg": st["exp_avg"].to(p.device, torch.float32).contiguous().clone(),
                    "exp_avg_sq": st["exp_avg_sq"].to(p.device, torch.float32).contiguous().clone()}
