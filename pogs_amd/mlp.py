"""Fused channel MLP on the rendered feature image (MI355X matrix cores).

Replaces, in the model's ``get_outputs`` (``pogs/pogs.py:516-522``), the alpha normalisation, the call of
``self.nn`` (four bias-free 64-wide ``Linear`` layers with ReLU between, ``pogs/pogs.py:153-161``) and the
eval-mode cut below alpha 0.8 with one kernel that reads the feature image once and writes the result once;
the intermediate activations never reach memory.  Arithmetic: split-bf16 by default (each float32 product as
three bf16 matrix products, error 2^-16 per product: 1e-4 of the output scale, the tolerance the rendered image
itself is held to); ``POGS_MLP_VARIANT=1`` selects the exact-float32 matrix instruction (summation order is the
only difference from torch).

``FeatureMLP`` is the reference module itself with another ``forward``: its ``state_dict`` keys are ``0.weight``,
``2.weight``, ``4.weight``, ``6.weight`` like the ``torch.nn.Sequential`` it stands in for, so checkpoints
load either way.

Gradients: with frozen weights (the tracker, ``pogs/tracking/rigid_group_optimizer.py``, optimises poses
through the render) the gradient with respect to the features is one more fused kernel that re-uses the ReLU
signs recorded by the forward.  When a weight needs a gradient (training: ``nn_projection`` is an optimiser group,
``pogs/pogs_config.py:78-85``) the forward also stores the three hidden activations and ONE backward kernel
(``pr_mlp_bwd_full``) produces the four weight gradients -- the reduction over pixels as matrix products with both
factors transposed through LDS -- and the feature gradient.  There is no CPU path and no torch GEMM.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import Tensor

from . import _lib
from .rendering import _call, _ptr, _stream

DIM = 64
LAYERS = 4


def _check(feats: Tensor, alphas: Optional[Tensor], weights: Sequence[Tensor]):
    if len(weights) != LAYERS:
        raise ValueError(f"the channel MLP has {LAYERS} weight matrices, got {len(weights)}")
    if not feats.is_cuda:
        raise RuntimeError("pogs_amd.mlp runs on a HIP device only (no CPU fallback)")
    if feats.shape[-1] != DIM:
        raise ValueError(f"feature images must have {DIM} channels, got {feats.shape[-1]}")
    for w in weights:
        if tuple(w.shape) != (DIM, DIM):
            raise ValueError(f"weights must be [{DIM}, {DIM}], got {tuple(w.shape)}")
        if w.device != feats.device:
            raise RuntimeError("weights and features are on different devices")
    if alphas is not None and alphas.numel() != feats.numel() // DIM:
        raise ValueError("alphas must hold one value per pixel")


def _f32(t: Tensor) -> Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class _FusedMLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, alphas, zero_below, w0, w1, w2, w3):
        x = _f32(feats).view(-1, DIM)
        al = None if alphas is None else _f32(alphas).view(-1)
        ws = [_f32(w) for w in (w0, w1, w2, w3)]
        P = x.shape[0]
        out = torch.empty_like(x)
        lib = _lib.load()
        ctx.weights_need_grad = any(ctx.needs_input_grad[3:])
        masks = acts = None
        with torch.cuda.device(x.device):
            if ctx.weights_need_grad:
                # training: the hidden activations are kept for the weight gradients
                acts = torch.empty(LAYERS - 1, P, DIM, dtype=torch.float32, device=x.device)
                _call("pr_mlp_fwd_train", P, _ptr(x), _ptr(al), *[_ptr(w) for w in ws], float(zero_below), _ptr(out),
                      _ptr(acts), _stream())
            else:
                if ctx.needs_input_grad[0]:
                    masks = torch.empty(lib.pr_mlp_mask_words(P), dtype=torch.int64, device=x.device)
                _call("pr_mlp_fwd", P, _ptr(x), _ptr(al), *[_ptr(w) for w in ws], float(zero_below), _ptr(out),
                      _ptr(masks), _stream())
        ctx.zero_below = float(zero_below)
        ctx.shape = feats.shape
        ctx.save_for_backward(x if ctx.weights_need_grad else None, al, masks, acts, *ws)
        return out.view(*feats.shape[:-1], DIM)

    @staticmethod
    def backward(ctx, v_out):
        x, al, masks, acts, *ws = ctx.saved_tensors
        g = _f32(v_out).view(-1, DIM)
        P = g.shape[0]
        v_feats = None
        grads_w = [None] * LAYERS
        with torch.cuda.device(g.device):
            if ctx.weights_need_grad:
                v_w = torch.zeros(LAYERS, DIM, DIM, dtype=torch.float32, device=g.device)
                v = torch.empty_like(g) if ctx.needs_input_grad[0] else None
                _call("pr_mlp_bwd_full", P, _ptr(g), _ptr(x), _ptr(al), _ptr(acts), *[_ptr(w) for w in ws],
                      ctx.zero_below, _ptr(v), _ptr(v_w), _stream())
                v_feats = None if v is None else v.view(ctx.shape)
                grads_w = [v_w[i] if ctx.needs_input_grad[3 + i] else None for i in range(LAYERS)]
            elif ctx.needs_input_grad[0]:
                v = torch.empty_like(g)
                _call("pr_mlp_bwd_input", P, _ptr(g), _ptr(al), *[_ptr(w) for w in ws], ctx.zero_below, _ptr(masks),
                      _ptr(v), _stream())
                v_feats = v.view(ctx.shape)
        return (v_feats, None, None, *grads_w)


def feature_mlp(feats: Tensor, alphas: Optional[Tensor], weights: Sequence[Tensor], zero_below: float = 0.0) -> Tensor:
    """``feats``: [..., 64] rendered feature image; ``alphas``: [..., 1] (or [...]) accumulated opacity or None
    (no normalisation); ``weights``: the four ``Linear.weight`` matrices; ``zero_below``: rows with alpha below it
    are zeroed (0.8 in the reference's eval mode, ``pogs/pogs.py:521-522``; 0 = off).  Returns [..., 64]."""
    _check(feats, alphas, weights)
    return _FusedMLP.apply(feats, alphas, zero_below, *weights)


class FeatureMLP(torch.nn.Sequential):
    """Drop-in for the model's ``self.nn`` (``pogs/pogs.py:153-161``) including what surrounds its call
    (``pogs/pogs.py:516-522``).  It IS the reference's ``Sequential`` (same submodules, initialisation and
    ``state_dict`` keys ``0.weight, 2.weight, 4.weight, 6.weight``); only ``forward`` differs:
    ``forward(feats, alphas)`` normalises by alpha, applies the layers and, outside training, the 0.8 cut."""

    EVAL_ALPHA_CUT = 0.8

    def __init__(self, device=None):
        layers = []
        for i in range(LAYERS):
            layers.append(torch.nn.Linear(DIM, DIM, bias=False, device=device))
            if i < LAYERS - 1:
                layers.append(torch.nn.ReLU())
        super().__init__(*layers)

    def linear_weights(self):
        return [m.weight for m in self if isinstance(m, torch.nn.Linear)]

    def forward(self, feats: Tensor, alphas: Optional[Tensor] = None) -> Tensor:
        cut = 0.0 if self.training or alphas is None else self.EVAL_ALPHA_CUT
        return feature_mlp(feats, alphas, self.linear_weights(), cut)
