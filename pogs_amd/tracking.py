"""Rigid-group transform of a Gaussian model on MI355X: the tracker's per-iteration pose update.

Mirrors the Warp kernel ``apply_to_model`` (``pogs/tracking/utils.py:53-93``) as the tracker launches it
(``pogs/tracking/rigid_group_optimizer.py:455-479``) and the ``wp.Tape`` backward that follows
``loss.backward()`` (``:413-430``): same argument order and meaning, and the gradient reaches
``p_deltas`` only.  Here the two steps are one autograd function, so ``loss.backward()`` alone fills
``p_deltas.grad``.  All device work goes through ``libpogs_rast.so``; there is no PyTorch fallback.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from . import _lib
from .rendering import _call, _ptr, _stream


def _check(init_p2ws, p_deltas, group_labels, means, quats):
    for t in (init_p2ws, p_deltas, group_labels, means, quats):
        if not t.is_cuda:
            raise RuntimeError("pogs_amd.tracking.apply_to_model needs tensors on a HIP device (no CPU fallback)")
    G, N = p_deltas.shape[0], means.shape[0]
    assert init_p2ws.shape == (G, 7), init_p2ws.shape
    assert p_deltas.shape == (G, 7), p_deltas.shape
    assert group_labels.shape == (N,), group_labels.shape
    assert means.shape == (N, 3), means.shape
    assert quats.shape == (N, 4), quats.shape
    for t in (init_p2ws, p_deltas, means, quats):
        if t.dtype != torch.float32:
            raise RuntimeError("pogs_amd.tracking.apply_to_model computes in float32")
    if group_labels.dtype != torch.int32:
        raise RuntimeError("group_labels must be int32 (the Warp kernel's `wp.array(dtype=int)`)")


class _ApplyToModel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, init_p2ws, p_deltas, group_labels, means, quats, quats_grad):
        _lib.load()
        init_p2ws, p_deltas = init_p2ws.contiguous(), p_deltas.contiguous()
        group_labels, means, quats = group_labels.contiguous(), means.contiguous(), quats.contiguous()
        N, G = means.shape[0], p_deltas.shape[0]
        means_out, quats_out = torch.empty_like(means), torch.empty_like(quats)
        _call("pr_rigid_fwd", N, G, _ptr(init_p2ws), _ptr(p_deltas), _ptr(group_labels), _ptr(means), _ptr(quats),
              _ptr(means_out), _ptr(quats_out), _stream())
        ctx.save_for_backward(init_p2ws, p_deltas, group_labels, means, quats)
        ctx.quats_grad = quats_grad
        if not quats_grad:
            ctx.mark_non_differentiable(quats_out)
        return means_out, quats_out

    @staticmethod
    def backward(ctx, v_means_out, v_quats_out):
        init_p2ws, p_deltas, group_labels, means, quats = ctx.saved_tensors
        v_p_deltas = None
        if ctx.needs_input_grad[1]:
            N, G = means.shape[0], p_deltas.shape[0]
            v_p_deltas = torch.zeros_like(p_deltas)
            v_means_out = v_means_out.contiguous()
            vq = v_quats_out.contiguous() if (ctx.quats_grad and v_quats_out is not None) else None
            _call("pr_rigid_bwd", N, G, _ptr(init_p2ws), _ptr(p_deltas), _ptr(group_labels), _ptr(means),
                  _ptr(quats), _ptr(v_means_out), _ptr(vq), _ptr(v_p_deltas), _stream())
        return None, v_p_deltas, None, None, None, None


def apply_to_model(init_p2ws: Tensor, p_deltas: Tensor, group_labels: Tensor, means: Tensor, quats: Tensor,
                   quats_grad: bool = False) -> Tuple[Tensor, Tensor]:
    """Move every Gaussian by its group's ``p2w * delta * inverse(p2w)``.

    init_p2ws [G,7], p_deltas [G,7]: poses as (x y z qw qx qy qz); group_labels [N] int32 in [0,G);
    means [N,3]; quats [N,4] (w x y z).  Returns (means_out [N,3], quats_out [N,4]).

    Only ``p_deltas`` receives a gradient.  With ``quats_grad=False`` (default) ``quats_out`` carries no
    gradient, as in the reference, which allocates it with ``requires_grad=False``
    (``rigid_group_optimizer.py:460-462``) so its tape only sees the cotangent of the means; pass
    ``quats_grad=True`` to differentiate through the rotations as well.
    """
    _check(init_p2ws, p_deltas, group_labels, means, quats)
    return _ApplyToModel.apply(init_p2ws, p_deltas, group_labels, means, quats, quats_grad)


def identity_7vec(device="cuda") -> Tensor:
    """The identity pose as a [1,7] row (pogs/tracking/utils.py:95-99)."""
    return torch.tensor([[0, 0, 0, 1, 0, 0, 0]], dtype=torch.float32, device=device)
