"""Rigid-group transform of a Gaussian model on MI355X: the tracker's per-iteration pose update.

Mirrors the Warp kernel ``apply_to_model`` (``pogs/tracking/utils.py:53-93``) as the tracker launches it
(``pogs/tracking/rigid_group_optimizer.py:455-479``) and the ``wp.Tape`` backward that follows
``loss.backward()`` (``:413-430``): same argument order and meaning, and the gradient reaches
``p_deltas`` only.  Here the two steps are one autograd function, so ``loss.backward()`` alone fills
``p_deltas.grad``.  All device work goes through ``libpogs_rast.so``; there is no PyTorch fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

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
        with torch.cuda.device(means.device):   # stream and launch device = the tensors' device, not the current one
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
            with torch.cuda.device(means.device):
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


class TrackingLoop:
    """The tracker's per-frame pose optimisation around the rasterizer, as ``RigidGroupOptimizer.step`` runs it
    (``pogs/tracking/rigid_group_optimizer.py:402-453`` with ``get_optim_loss`` ``:278-400`` on the whole-frame
    branch ``:298-307``): per iteration renormalise the rotation part of the pose deltas (``:411-412``), move every
    Gaussian by its group's delta (``apply_to_model``), render RGB + expected depth (SH degree 3, tile 8:
    ``get_outputs(..., BLOCK_WIDTH=8)``) and the 64-channel feature image (tile 16, everything but the means
    detached: ``pogs/pogs.py:493-510``) pushed through the frozen channel MLP (``pogs/pogs.py:516-520``), take the
    feature loss ``(real - rendered).norm(dim=-1).mean()`` (``:364``) plus, optionally, the clamped depth term
    (``:373-386``), back-propagate to the ``[G,7]`` deltas and take an Adam step whose rate decays exponentially from
    ``pose_lr`` to ``pose_lr_final`` over the frame's iterations (``:403-407``, config ``:26-45``).  The reference's
    Gaussian blur of the rendered features (``:304``) is left to the caller: it is not part of this path.

    Everything on one HIP device; appearance is frozen, so the compositor's backward runs without colour gradients.
    """

    def __init__(self, means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, sh: Tensor, features: Tensor,
                 group_labels: Tensor, n_groups: int, viewmat: Tensor, K: Tensor, width: int, height: int, mlp=None,
                 pose_lr: float = 0.004, pose_lr_final: float = 0.0008, depth_loss_mult: float = 3.7,
                 dataset_scale: float = 1.0, fused_transform: bool = True):
        from .mlp import FeatureMLP

        dev = means.device
        self.means, self.quats = means.detach().contiguous(), quats.detach().contiguous()
        self.scales, self.opacities = scales.detach().contiguous(), opacities.detach().contiguous()
        self.sh, self.features = sh.detach().contiguous(), features.detach().contiguous()
        self.labels = group_labels.to(torch.int32).contiguous()
        self.viewmat, self.K, self.width, self.height = viewmat, K, int(width), int(height)
        self.net = mlp if mlp is not None else FeatureMLP(device=dev)
        for prm in self.net.parameters():
            prm.requires_grad_(False)
        self.net.train()        # no alpha cut inside the optimisation
        self.pose_lr, self.pose_lr_final = pose_lr, pose_lr_final
        self.depth_loss_mult, self.dataset_scale = depth_loss_mult, dataset_scale
        # True: the group transform runs inside the projection kernels (rasterization(rigid=...)); False: as its own
        # pass over the model (apply_to_model), the reference's structure.  Same results either way.
        self.fused_transform = fused_transform
        # part-to-world poses: identity rotation at each group's centroid (rigid_group_optimizer.py:108-118)
        self.init_p2w = torch.zeros(n_groups, 7, device=dev)
        self.init_p2w[:, 3] = 1.0
        for g in range(n_groups):
            self.init_p2w[g, :3] = self.means[self.labels == g].mean(dim=0)
        self.part_deltas = torch.zeros(n_groups, 7, device=dev)
        self.part_deltas[:, 3] = 1.0
        self.part_deltas.requires_grad_(True)
        self.optimizer = torch.optim.Adam([self.part_deltas], lr=pose_lr)
        self.target = None

    @classmethod
    def synthetic(cls, dev, n=200_000, width=640, height=360, groups=4, seed=0, offset=0.02):
        """A seeded scene of ``n`` Gaussians in ``groups`` rigid groups (angular sectors around the view axis, like
        separate objects), with features that vary smoothly over space (as learned DINO features do over a surface),
        observed after every group was moved by a small pose offset: the loop has something to recover."""
        import math

        from .scenes import make_scene

        sc = make_scene(n, width, height, n_features=64, seed=seed)
        g = torch.Generator().manual_seed(seed + 1)
        angle = torch.atan2(sc.means[:, 1], sc.means[:, 0]) + math.pi
        labels = torch.clamp((angle / (2 * math.pi) * groups).long(), max=groups - 1)
        feats = torch.sin(sc.means @ (torch.randn(3, 64, generator=g) * 2.0) + torch.rand(64, generator=g) * 6.28)
        from .mlp import FeatureMLP

        with torch.random.fork_rng(devices=[dev]):   # the channel MLP's random weights are part of the seeded scene
            torch.manual_seed(seed + 2)
            net = FeatureMLP(device=dev)
        self = cls(sc.means.to(dev), torch.nn.functional.normalize(sc.quats, dim=-1).to(dev),
                   sc.log_scales.exp().to(dev), torch.sigmoid(sc.opacity_logits).to(dev), sc.sh.to(dev),
                   feats.to(dev), labels.to(dev), groups, sc.viewmats[:1].to(dev), sc.Ks[:1].to(dev), width, height,
                   mlp=net)
        moved = torch.zeros(groups, 7)
        moved[:, :3] = torch.randn(groups, 3, generator=g) * offset
        moved[:, 3] = 1.0
        moved[:, 4:] = torch.randn(groups, 3, generator=g) * offset
        moved[:, 3:] = torch.nn.functional.normalize(moved[:, 3:], dim=-1)
        with torch.no_grad():
            self.set_observation(*self.render(moved.to(dev))[:2])
        return self

    def render(self, deltas: Tensor):
        """-> (features after the channel MLP [1,H,W,64], expected depth [1,H,W,1], rgb [1,H,W,3])."""
        from .rendering import rasterization

        common = dict(viewmats=self.viewmat, Ks=self.K, width=self.width, height=self.height, packed=False)
        if self.fused_transform:
            m, q = self.means, self.quats
            common["rigid"] = (self.labels, self.init_p2w, deltas)
        else:
            m, q = apply_to_model(self.init_p2w, deltas, self.labels, self.means, self.quats)
        rgbd, _, _ = rasterization(m, q, self.scales, self.opacities, self.sh, sh_degree=3, render_mode="RGB+ED",
                                   tile_size=8, **common)
        f, fa, _ = rasterization(m, q.detach(), self.scales, self.opacities, self.features, **common)
        return self.net(f, fa), rgbd[..., 3:4], rgbd[..., :3]

    def set_observation(self, dino: Tensor, depth: Optional[Tensor] = None) -> None:
        self.target = (dino.detach(), None if depth is None else depth.detach())

    def loss(self, use_depth: bool = True) -> Tensor:
        dino, depth, _ = self.render(self.part_deltas)
        loss = (self.target[0] - dino).norm(dim=-1).nanmean()
        if use_depth and self.target[1] is not None:
            a = torch.clamp(depth / self.dataset_scale, min=1e-8, max=1.0)
            b = torch.clamp(self.target[1] / self.dataset_scale, min=1e-8, max=1.0)
            loss = loss + self.depth_loss_mult * ((a - b) ** 2).mean()
        return loss

    def frame(self, niter: int = 50, use_depth: bool = True):
        """One frame's ``niter`` iterations; returns the loss of every iteration (read back once, at the end)."""
        assert self.target is not None, "set_observation first"
        gamma = (self.pose_lr_final / self.pose_lr) ** (1.0 / max(niter, 1))
        losses = []
        for i in range(niter):
            with torch.no_grad():
                self.part_deltas[:, 3:] = self.part_deltas[:, 3:] / self.part_deltas[:, 3:].norm(dim=1, keepdim=True)
            self.optimizer.zero_grad(set_to_none=True)
            loss = self.loss(use_depth)
            loss.backward()
            self.optimizer.step()
            self.optimizer.param_groups[0]["lr"] = self.pose_lr * gamma ** (i + 1)
            losses.append(loss.detach())
        self.optimizer.param_groups[0]["lr"] = self.pose_lr      # "reset lr" (:441)
        return torch.stack(losses).tolist()
