"""Densification strategy of the trainer on MI355X: grow (duplicate / split) and prune the Gaussian set.

Mirrors gsplat 1.4.0's ``DefaultStrategy`` (third-party ``gsplat/strategy/default.py`` with the operations of
``gsplat/strategy/ops.py``) as the reference constructs and drives it:
``nerfstudio/models/splatfacto.py:264-281`` (fields), ``pogs/pogs.py:373-379`` (``step_pre_backward`` after the
render), ``nerfstudio/models/splatfacto.py:365-375`` (``step_post_backward`` after ``loss.backward()``).
Same field names, method names, arguments and state keys (``grad2d``, ``count``, ``radii``, ``scene_scale``).

What differs is how the work is issued.  The per-step statistics are one kernel (``pr_densify_accumulate``).
A refinement classifies all Gaussians and scans the classes once (``pr_densify_plan``), reads four counters back
(the only host synchronisation), builds ONE gather map in the order ``duplicate -> split -> remove`` would leave
(``pr_densify_emit``) and then builds every parameter and optimiser-moment tensor with a single gather
(``pr_densify_gather`` / ``pr_densify_split_children``) instead of materialising each of them three times.
The normal samples of a split come from ``torch.randn(2, n_split, 3, device=...)`` -- the same draw the published
code makes, so a seeded run consumes the generator identically.

``optimizers`` may be the reference's ``{name: torch.optim.Optimizer}`` (one parameter per optimiser) or a
``pogs_amd.optim.Optimizers``.  HIP device tensors only; there is no CPU path.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .rendering import _call, _ptr, _stream


def _moment_tensors(optimizers, name: str, param: Tensor):
    """The optimiser-state dict of `param` (possibly empty) -- for either optimiser container."""
    from .optim import Optimizers
    if isinstance(optimizers, Optimizers):
        if name not in optimizers.parameters:
            return None
        for p, st in zip(optimizers.parameters[name], optimizers._state[name]):
            if p is param:
                return st
        return None
    if name not in optimizers:
        return None
    return optimizers[name].state.get(param, {})


def _swap_parameter(optimizers, name: str, old: Tensor, new: Tensor, new_state: Optional[dict]) -> None:
    from .optim import Optimizers
    if isinstance(optimizers, Optimizers):
        if name in optimizers.parameters:
            for k, p in enumerate(optimizers.parameters[name]):
                if p is old:
                    optimizers.parameters[name][k] = new
                    optimizers._state[name][k] = new_state if new_state is not None else {}
        return
    if name not in optimizers:
        assert not old.requires_grad, f"parameter '{name}' requires a gradient but has no optimizer"
        return
    opt = optimizers[name]
    if old in opt.state:
        del opt.state[old]
    for group in opt.param_groups:
        group["params"] = [new]
    if new_state:
        opt.state[new] = new_state


@dataclass
class DefaultStrategy:
    prune_opa: float = 0.005
    grow_grad2d: float = 0.0002
    grow_scale3d: float = 0.01
    grow_scale2d: float = 0.05
    prune_scale3d: float = 0.1
    prune_scale2d: float = 0.15
    refine_scale2d_stop_iter: int = 0
    refine_start_iter: int = 500
    refine_stop_iter: int = 15_000
    reset_every: int = 3000
    refine_every: int = 100
    pause_refine_after_reset: int = 0
    absgrad: bool = False
    revised_opacity: bool = False
    verbose: bool = False
    key_for_gradient: str = "means2d"

    # ---- interface of the published class -----------------------------------------------------------------
    def initialize_state(self, scene_scale: float = 1.0) -> Dict[str, Any]:
        state: Dict[str, Any] = {"grad2d": None, "count": None, "scene_scale": scene_scale}
        if self.refine_scale2d_stop_iter > 0:
            state["radii"] = None
        return state

    def check_sanity(self, params, optimizers) -> None:
        for key in ("means", "scales", "quats", "opacities"):
            assert key in params, f"{key} is required in params but missing."

    def step_pre_backward(self, params, optimizers, state, step: int, info: Dict[str, Any]) -> None:
        assert self.key_for_gradient in info, "The 2D means of the Gaussians is required but missing."
        info[self.key_for_gradient].retain_grad()

    def step_post_backward(self, params, optimizers, state, step: int, info: Dict[str, Any],
                           packed: bool = False) -> None:
        if step >= self.refine_stop_iter:
            return
        self._update_state(params, state, info, packed=packed)
        if (step > self.refine_start_iter and step % self.refine_every == 0
                and step % self.reset_every >= self.pause_refine_after_reset):
            n_dupli, n_split, n_prune = self._refine(params, optimizers, state, step)
            if self.verbose:
                n = len(params["means"])
                print(f"Step {step}: {n_dupli} GSs duplicated, {n_split} GSs split, {n_prune} GSs pruned. "
                      f"Now having {n} GSs.")
            state["grad2d"].zero_()
            state["count"].zero_()
            if self.refine_scale2d_stop_iter > 0:
                state["radii"].zero_()
        if step % self.reset_every == 0:
            self._reset_opa(params, optimizers, value=self.prune_opa * 2.0)

    # ---- per-step statistics ------------------------------------------------------------------------------
    def _update_state(self, params, state, info: Dict[str, Any], packed: bool = False) -> None:
        if packed:
            raise NotImplementedError("packed=True is not supported (the reference calls with packed=False)")
        for key in ("width", "height", "n_cameras", "radii", self.key_for_gradient):
            assert key in info, f"{key} is required but missing."
        m2d = info[self.key_for_gradient]
        grads = m2d.absgrad if self.absgrad else m2d.grad
        if grads is None:
            raise RuntimeError("the 2D means carry no gradient: call step_pre_backward before loss.backward()")
        if not grads.is_cuda:
            raise RuntimeError("pogs_amd.strategy runs on a HIP device only (no CPU fallback)")
        grads = grads.detach()
        if grads.dim() == 2:
            grads = grads[None]
        if grads.dtype != torch.float32 or not grads.is_contiguous():
            grads = grads.float().contiguous()
        radii = info["radii"]
        if radii.dim() == 1:
            radii = radii[None]
        if radii.dtype != torch.int32 or not radii.is_contiguous():
            radii = radii.to(torch.int32).contiguous()
        C, N = radii.shape
        n_gaussian = len(next(iter(params.values())))
        assert N == n_gaussian and tuple(grads.shape) == (C, N, 2)
        dev = grads.device
        if state["grad2d"] is None:
            state["grad2d"] = torch.zeros(n_gaussian, device=dev)
        if state["count"] is None:
            state["count"] = torch.zeros(n_gaussian, device=dev)
        track_radii = self.refine_scale2d_stop_iter > 0
        if track_radii and state["radii"] is None:
            state["radii"] = torch.zeros(n_gaussian, device=dev)
        with torch.cuda.device(dev):
            _call("pr_densify_accumulate", C, N, _ptr(grads), _ptr(radii), int(info["width"]), int(info["height"]),
                  int(info["n_cameras"]), _ptr(state["grad2d"]), _ptr(state["count"]),
                  _ptr(state["radii"]) if track_radii else None, _stream())

    # ---- refinement ---------------------------------------------------------------------------------------
    def _rule(self, state, step: int) -> "_lib.DensifyRule":
        scale = float(state["scene_scale"])
        by_radius = step < self.refine_scale2d_stop_iter
        big = step > self.reset_every
        return _lib.DensifyRule(self.grow_grad2d, self.grow_scale3d * scale, self.grow_scale2d, self.prune_opa,
                                self.prune_scale3d * scale, self.prune_scale2d, int(by_radius), int(big),
                                int(big and by_radius), int(self.revised_opacity))

    @torch.no_grad()
    def _refine(self, params, optimizers, state, step: int) -> Tuple[int, int, int]:
        """_grow_gs followed by _prune_gs of the published class, as one gather."""
        lib = _lib.load()
        means = params["means"]
        N, dev = len(means), means.device
        scales, quats, opac = params["scales"], params["quats"], params["opacities"]
        for t in (means, scales, quats, opac):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("pogs_amd.strategy needs contiguous float32 parameters on a HIP device")
        if opac.numel() != N:
            raise ValueError("opacities must hold one value per Gaussian")
        rule = self._rule(state, step)
        radii_max = state.get("radii") if (rule.grow_by_radius or rule.prune_by_radius) else None
        scratch_bytes = lib.pr_densify_plan_scratch_bytes(N)
        scratch = torch.empty(max(scratch_bytes, 1), dtype=torch.uint8, device=dev)
        counts = torch.empty(4, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _call("pr_densify_plan", N, _ptr(state["grad2d"]), _ptr(state["count"]), _ptr(scales), _ptr(opac),
                  _ptr(radii_max), ctypes.byref(rule), _ptr(counts), _ptr(scratch), scratch_bytes, _stream())
            n_out, n_dupli, n_split, n_prune = (int(v) for v in counts.cpu())       # the one host wait
            if n_dupli == 0 and n_split == 0 and n_prune == 0:
                return 0, 0, 0
            map_src = torch.empty(max(n_out, 1), dtype=torch.int32, device=dev)
            map_tag = torch.empty(max(n_out, 1), dtype=torch.int32, device=dev)
            _call("pr_densify_emit", N, _ptr(scratch), scratch_bytes, n_out, _ptr(map_src), _ptr(map_tag), _stream())

            def gather(src: Tensor, zero_new: bool) -> Tensor:
                out = torch.empty((n_out, *src.shape[1:]), dtype=src.dtype, device=dev)
                if n_out == 0:
                    return out
                if src.dtype == torch.float32 and src.is_contiguous():
                    width = src[0].numel() if src.dim() > 1 else 1
                    _call("pr_densify_gather", n_out, width, _ptr(src), _ptr(map_src), _ptr(map_tag), int(zero_new),
                          _ptr(out), _stream())
                else:   # another dtype: the same gather with device index operators
                    out = src[map_src[:n_out].long()]
                    if zero_new:
                        out[(map_tag[:n_out] >> 30) != 0] = 0
                return out

            new = {name: gather(p.detach(), False) for name, p in params.items()}
            if n_split > 0 and n_out > 0:
                noise = torch.randn(2, n_split, 3, device=dev)
                revised = bool(self.revised_opacity)
                _call("pr_densify_split_children", n_out, _ptr(map_src), _ptr(map_tag), n_split, _ptr(means),
                      _ptr(scales), _ptr(quats), _ptr(opac) if revised else None, _ptr(noise), _ptr(new["means"]),
                      _ptr(new["scales"]), _ptr(new["opacities"]) if revised else None, _stream())
            for name in list(params.keys()):
                old = params[name]
                moments = _moment_tensors(optimizers, name, old)
                new_state = None
                if moments:
                    new_state = {k: (gather(v, True) if isinstance(v, Tensor) and v.dim() > 0 and len(v) == N else v)
                                 for k, v in moments.items()}
                fresh = torch.nn.Parameter(new[name], requires_grad=old.requires_grad)
                params[name] = fresh
                _swap_parameter(optimizers, name, old, fresh, new_state)
            for k, v in list(state.items()):
                if isinstance(v, Tensor):
                    state[k] = gather(v, False)
        return n_dupli, n_split, n_prune

    @torch.no_grad()
    def _reset_opa(self, params, optimizers, value: float) -> None:
        """ops.py reset_opa: clamp the opacity parameter to logit(value) and zero its optimiser moments."""
        old = params["opacities"]
        cap = torch.logit(torch.tensor(value)).item()
        fresh = torch.nn.Parameter(torch.clamp(old.detach(), max=cap), requires_grad=old.requires_grad)
        moments = _moment_tensors(optimizers, "opacities", old)
        new_state = None
        if moments:
            new_state = {k: (torch.zeros_like(v) if isinstance(v, Tensor) and k != "step" else v)
                         for k, v in moments.items()}
        params["opacities"] = fresh
        _swap_parameter(optimizers, "opacities", old, fresh, new_state)
