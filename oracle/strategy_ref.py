"""TEST INFRASTRUCTURE ONLY -- CPU restatement of gsplat 1.4.0's ``DefaultStrategy`` (densify / prune).

PARITY UNPINNED: the algorithm lives in the third-party dependency ``gsplat`` (1.4.0; files
``gsplat/strategy/default.py`` and ``gsplat/strategy/ops.py``), which is not vendored in the reference tree, and
the reference holds no fixtures for it.  This file restates the published algorithm; parity is anchored on the
reference's call sites: construction ``nerfstudio/models/splatfacto.py:264-281``, ``step_pre_backward``
``pogs/pogs.py:373-379``, ``step_post_backward`` ``nerfstudio/models/splatfacto.py:365-375``.

Written with plain torch CPU operators, one step at a time (accumulate -> duplicate -> split -> prune ->
opacity reset), each of them materialising the whole parameter set the way the published code does.

Only tests/ and bench scripts' CPU legs may import this module.
"""
import math
from dataclasses import dataclass

import torch


@dataclass
class Config:
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


def initialize_state(cfg, scene_scale=1.0):
    state = {"grad2d": None, "count": None, "scene_scale": scene_scale}
    if cfg.refine_scale2d_stop_iter > 0:
        state["radii"] = None
    return state


def update_state(cfg, state, n_gaussians, grads, radii, width, height, n_cameras):
    """grads [C, N, 2] (the 2D-mean gradient or its absolute-value variant), radii [C, N] integer pixels."""
    g = grads.clone()
    g[..., 0] *= width / 2.0 * n_cameras
    g[..., 1] *= height / 2.0 * n_cameras
    if state["grad2d"] is None:
        state["grad2d"] = torch.zeros(n_gaussians)
    if state["count"] is None:
        state["count"] = torch.zeros(n_gaussians)
    if cfg.refine_scale2d_stop_iter > 0 and state["radii"] is None:
        state["radii"] = torch.zeros(n_gaussians)
    visible = radii > 0.0
    ids = torch.where(visible)[1]
    state["grad2d"].index_add_(0, ids, g[visible].norm(dim=-1))
    state["count"].index_add_(0, ids, torch.ones_like(ids, dtype=torch.float32))
    if cfg.refine_scale2d_stop_iter > 0:
        # the published code writes max(old, new) per (camera, Gaussian) pair; with one camera that IS the maximum,
        # with several the intended scatter-max is what is restated here
        norm = radii.float() / float(max(width, height))
        for c in range(radii.shape[0]):
            v = visible[c]
            state["radii"][v] = torch.maximum(state["radii"][v], norm[c][v])


def _rotmat(q):
    w, x, y, z = torch.unbind(q, dim=-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def _rebuild(params, adam, fn_param, fn_state, names=None):
    """params: {name: tensor}; adam: {name: {"exp_avg": t, "exp_avg_sq": t}} (missing name = no state yet)."""
    for name in (list(params) if names is None else names):
        params[name] = fn_param(name, params[name])
        if name in adam:
            for key in ("exp_avg", "exp_avg_sq"):
                adam[name][key] = fn_state(adam[name][key])


def duplicate(params, adam, state, mask):
    sel = torch.where(mask)[0]
    _rebuild(params, adam, lambda n, p: torch.cat([p, p[sel]]),
             lambda v: torch.cat([v, torch.zeros((len(sel), *v.shape[1:]))]))
    for k, v in state.items():
        if isinstance(v, torch.Tensor):
            state[k] = torch.cat([v, v[sel]])


def split(params, adam, state, mask, noise, revised_opacity=False):
    """noise: [2, n_split, 3] standard-normal samples (``torch.randn(2, n, 3)`` in the published code)."""
    sel, rest = torch.where(mask)[0], torch.where(~mask)[0]
    scales = torch.exp(params["scales"][sel])
    rot = _rotmat(torch.nn.functional.normalize(params["quats"][sel], dim=-1))
    offsets = torch.einsum("nij,nj,bnj->bni", rot, scales, noise)

    def fn_param(name, p):
        twice = [2] + [1] * (p.dim() - 1)
        if name == "means":
            kids = (p[sel] + offsets).reshape(-1, 3)
        elif name == "scales":
            kids = torch.log(scales / 1.6).repeat(2, 1)
        elif name == "opacities" and revised_opacity:
            kids = torch.logit(1.0 - torch.sqrt(1.0 - torch.sigmoid(p[sel]))).repeat(twice)
        else:
            kids = p[sel].repeat(twice)
        return torch.cat([p[rest], kids])

    _rebuild(params, adam, fn_param, lambda v: torch.cat([v[rest], torch.zeros((2 * len(sel), *v.shape[1:]))]))
    for k, v in state.items():
        if isinstance(v, torch.Tensor):
            state[k] = torch.cat([v[rest], v[sel].repeat([2] + [1] * (v.dim() - 1))])


def remove(params, adam, state, mask):
    keep = torch.where(~mask)[0]
    _rebuild(params, adam, lambda n, p: p[keep], lambda v: v[keep])
    for k, v in state.items():
        if isinstance(v, torch.Tensor):
            state[k] = v[keep]


def reset_opa(params, adam, value):
    cap = torch.logit(torch.tensor(value)).item()
    _rebuild(params, adam, lambda n, p: torch.clamp(p, max=cap), torch.zeros_like, names=["opacities"])


def grow(cfg, params, adam, state, step, noise_fn):
    """Returns (n_duplicated, n_split).  noise_fn(n) -> [2, n, 3]."""
    mean_grad = state["grad2d"] / state["count"].clamp_min(1)
    high = mean_grad > cfg.grow_grad2d
    small = torch.exp(params["scales"]).max(dim=-1).values <= cfg.grow_scale3d * state["scene_scale"]
    is_dup = high & small
    is_split = high & ~small
    if step < cfg.refine_scale2d_stop_iter:
        is_split |= state["radii"] > cfg.grow_scale2d
    n_dup, n_split = int(is_dup.sum()), int(is_split.sum())
    if n_dup > 0:
        duplicate(params, adam, state, is_dup)
    is_split = torch.cat([is_split, torch.zeros(n_dup, dtype=torch.bool)])
    if n_split > 0:
        split(params, adam, state, is_split, noise_fn(n_split), cfg.revised_opacity)
    return n_dup, n_split


def prune(cfg, params, adam, state, step):
    gone = torch.sigmoid(params["opacities"].flatten()) < cfg.prune_opa
    if step > cfg.reset_every:
        big = torch.exp(params["scales"]).max(dim=-1).values > cfg.prune_scale3d * state["scene_scale"]
        if step < cfg.refine_scale2d_stop_iter:
            big |= state["radii"] > cfg.prune_scale2d
        gone = gone | big
    n = int(gone.sum())
    if n > 0:
        remove(params, adam, state, gone)
    return n


def step_post_backward(cfg, params, adam, state, step, grads, radii, width, height, n_cameras, noise_fn):
    """One call of the strategy after the backward pass.  Returns (n_dup, n_split, n_pruned) or None when no
    refinement happened at this step."""
    if step >= cfg.refine_stop_iter:
        return None
    update_state(cfg, state, len(next(iter(params.values()))), grads, radii, width, height, n_cameras)
    done = None
    if (step > cfg.refine_start_iter and step % cfg.refine_every == 0
            and step % cfg.reset_every >= cfg.pause_refine_after_reset):
        n_dup, n_split = grow(cfg, params, adam, state, step, noise_fn)
        n_pruned = prune(cfg, params, adam, state, step)
        state["grad2d"].zero_()
        state["count"].zero_()
        if cfg.refine_scale2d_stop_iter > 0:
            state["radii"].zero_()
        done = (n_dup, n_split, n_pruned)
    if step % cfg.reset_every == 0:
        reset_opa(params, adam, cfg.prune_opa * 2.0)
    return done


LOGIT = lambda v: math.log(v / (1.0 - v))  # noqa: E731
