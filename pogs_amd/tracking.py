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


class _TrackLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rendered, target, depth, target_depth, valid, inv_scale, cut2, mult):
        _lib.load()
        D = rendered.shape[-1]
        rendered, target = rendered.contiguous().view(-1, D), target.contiguous().view(-1, D)
        P = rendered.shape[0]
        assert target.shape == rendered.shape and rendered.dtype == target.dtype == torch.float32
        if depth is not None:
            depth, target_depth = depth.contiguous().view(-1), target_depth.contiguous().view(-1)
            assert depth.shape == target_depth.shape == (P,)
        if valid is not None:
            valid = valid.contiguous().view(-1)
            assert valid.dtype == torch.bool and valid.shape == (P,)
        sums = torch.empty(4, dtype=torch.float32, device=rendered.device)
        loss = torch.empty((), dtype=torch.float32, device=rendered.device)
        with torch.cuda.device(rendered.device):
            _call("pr_track_loss_fwd", P, D, _ptr(rendered), _ptr(target), _ptr(depth), _ptr(target_depth), _ptr(valid),
                  inv_scale, cut2, mult, _ptr(sums), _ptr(loss), _stream())
        ctx.save_for_backward(rendered, target, depth, target_depth, valid, sums)
        ctx.consts = (inv_scale, cut2, mult)
        return loss

    @staticmethod
    def backward(ctx, v_loss):
        rendered, target, depth, target_depth, valid, sums = ctx.saved_tensors
        inv_scale, cut2, mult = ctx.consts
        P, D = rendered.shape
        v_rendered = torch.empty_like(rendered)
        v_depth = torch.empty_like(depth) if depth is not None and ctx.needs_input_grad[2] else None
        v_loss = v_loss.contiguous().float()
        with torch.cuda.device(rendered.device):
            _call("pr_track_loss_bwd", P, D, _ptr(rendered), _ptr(target), _ptr(depth), _ptr(target_depth), _ptr(valid),
                  inv_scale, cut2, mult, _ptr(sums), _ptr(v_loss), _ptr(v_rendered), _ptr(v_depth), _stream())
        return v_rendered, None, v_depth, None, None, None, None, None


def tracking_loss(rendered: Tensor, target: Tensor, depth: Optional[Tensor] = None, target_depth: Optional[Tensor] = None,
                  valid: Optional[Tensor] = None, dataset_scale: float = 1.0, depth_ignore_threshold: float = 0.26,
                  depth_loss_mult: float = 3.7) -> Tensor:
    """The tracker's loss (``rigid_group_optimizer.py:364, 373-386``) in one kernel, with one more for its gradients:
    ``(target - rendered).norm(dim=-1).nanmean()`` over the [..., D] features plus, when depths are given,
    ``depth_loss_mult`` times the mean over the kept pixels of ``(clamp(depth / dataset_scale, 1e-8, 1) -
    clamp(target_depth, 1e-8, 1)) ** 2`` -- kept: ``valid`` (bool, default all) and squared error below
    ``depth_ignore_threshold ** 2``.  Gradients reach ``rendered`` and ``depth``.  The returned tensor keeps the
    shapes given: the gradient of ``rendered`` comes back in its own shape."""
    shape, dshape = rendered.shape, None if depth is None else depth.shape
    out = _TrackLoss.apply(rendered.reshape(-1, shape[-1]), target.reshape(-1, shape[-1]),
                           None if depth is None else depth.reshape(-1),
                           None if target_depth is None else target_depth.reshape(-1),
                           None if valid is None else valid.reshape(-1), 1.0 / dataset_scale,
                           depth_ignore_threshold ** 2, depth_loss_mult)
    del dshape
    return out


def identity_7vec(device="cuda") -> Tensor:
    """The identity pose as a [1,7] row (pogs/tracking/utils.py:95-99)."""
    return torch.tensor([[0, 0, 0, 1, 0, 0, 0]], dtype=torch.float32, device=device)


@torch.no_grad()
def roi_depth_masks(acc: Tensor, observed_depth: Tensor, sizes: Tensor, own: Optional[Tensor] = None,
                    acc_min: float = 0.85, reject_factor: float = 1.3):
    """Depth masks of the per-object crops (``rigid_group_optimizer.py:316-337``) for all crops of a pass in four
    launches (``pr_roi_depth_masks``).  ``acc`` (rendered accumulation) and ``observed_depth`` (NaN where unobserved):
    [G,Hm,Wm]; crop g is the top-left ``sizes[g] = (h, w)`` window of image g.  -> (keep [G,Hm,Wm] bool, the observed
    depth where kept, else 0).  ``own`` is not needed (the windows come from ``sizes``)."""
    if not acc.is_cuda:
        raise RuntimeError("pogs_amd.tracking needs tensors on a HIP device: there is no CPU fallback")
    G, Hm, Wm = acc.shape
    acc, observed_depth = acc.contiguous().float(), observed_depth.contiguous().float()
    assert observed_depth.shape == acc.shape and sizes.shape == (G, 2) and sizes.dtype == torch.int32
    dev = acc.device
    scratch = torch.empty(_lib.load().pr_roi_depth_masks_scratch_bytes(G, Hm, Wm), dtype=torch.uint8, device=dev)
    medians = torch.empty(G, dtype=torch.float32, device=dev)
    keep = torch.empty(G, Hm, Wm, dtype=torch.uint8, device=dev)
    masked = torch.empty(G, Hm, Wm, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _call("pr_roi_depth_masks", G, Hm, Wm, _ptr(sizes.contiguous()), _ptr(acc), _ptr(observed_depth), float(acc_min),
              float(reject_factor), _ptr(scratch), _ptr(medians), _ptr(keep), _ptr(masked), _stream())
    return keep.view(torch.bool), masked


@torch.no_grad()
def roi_depth_masks_torch(acc: Tensor, observed_depth: Tensor, sizes: Tensor, own: Tensor, acc_min: float = 0.85,
                          reject_factor: float = 1.3):
    """The same masks from PyTorch operators (about forty launches and a sort-based median): what the fused kernels
    are tested against.  ``own`` [G,Hm,Wm] bool: each crop's own pixels."""
    G = own.shape[0]
    t_depth = observed_depth
    valid = (acc > acc_min) & ~t_depth.isnan()

    # erosions see the pixels beyond a crop as set, as the crop's own border is seen when it is processed alone
    def erode(m, k):
        x = (m | ~own)[:, None].float()
        return (-torch.nn.functional.max_pool2d(-x, k, stride=1, padding=k // 2))[:, 0] > 0.5

    valid_depths = erode(valid, 5) & own
    med = torch.where(valid_depths, t_depth, torch.full_like(t_depth, float("nan"))).view(G, -1).nanmedian(dim=1).values
    reject = t_depth > (med * reject_factor)[:, None, None]     # (a crop without a valid depth: nan, rejects nothing)
    valid_depths = valid_depths & ~reject
    masked = torch.where(valid_depths, t_depth, torch.zeros_like(t_depth))
    keep = erode(valid_depths & (masked != 0), 9) & own
    return keep, torch.where(keep, masked, torch.zeros_like(masked))


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

    ``set_rois`` switches to the per-object branch (``use_roi``, the reference config's default, ``:42`` and
    ``:309-364``): one crop camera per rigid group (``observation.py:157-174``: the box in pixels, sides cut to
    multiples of the 14-pixel DINO patch, principal point shifted by the corner, rescaled when a side exceeds 490
    pixels), and per iteration, for every group, a render of THAT group's Gaussians alone (``obj_id`` gather,
    ``pogs/pogs.py:287-300``) through its crop camera; the losses run over the concatenated pixels of all crops, the
    depth term over pixels that are covered (accumulation > 0.85), observed, not depth outliers and survive the 5x5 and
    9x9 erosions (``:317-337``) and whose squared error stays under ``depth_ignore_threshold``^2 (``:384``).
    """

    MAX_ROI_RESOLUTION = 490   # observation.py:117
    DINO_PATCH = 14

    def __init__(self, means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, sh: Tensor, features: Tensor,
                 group_labels: Tensor, n_groups: int, viewmat: Tensor, K: Tensor, width: int, height: int, mlp=None,
                 pose_lr: float = 0.004, pose_lr_final: float = 0.0008, depth_loss_mult: float = 3.7,
                 dataset_scale: float = 1.0, fused_transform: bool = True, depth_ignore_threshold: float = 0.26):
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
        self.depth_ignore_threshold = depth_ignore_threshold
        self.rois = None       # per group: crop camera, Gaussian subset, observation
        # per-object branch: all crops in one render call per pass (rasterization(camera_groups=...)) instead of one
        # call per object; False keeps the reference's structure (a loop over objects on gathered subsets)
        self.batch_rois = True
        # True: the group transform runs inside the projection kernels (rasterization(rigid=...)); False: as its own
        # pass over the model (apply_to_model), the reference's structure.  Same results either way.
        self.fused_transform = fused_transform
        # True: both renders of an iteration at tile 16, the second over the first one's per-tile lists
        # (rasterization(binning=...)); False: tile 8 then tile 16 as the reference, sharing the depth order only
        self.share_binning = True
        self.skip_unused_rgb = False
        # per-object branch: the crops' depth masks from pr_roi_depth_masks (False: from PyTorch operators)
        self.fused_roi_masks = True
        self.shared_tile = 16       # (8 for both measured the same: 57.5 against 56.7 ms per frame)
        # True: RGB+ED and the features as ONE render of 68 channels (whole-frame branch)
        self.fuse_renders = False
        # part-to-world poses: identity rotation at each group's centroid (rigid_group_optimizer.py:108-118)
        self.init_p2w = torch.zeros(n_groups, 7, device=dev)
        self.init_p2w[:, 3] = 1.0
        for g in range(n_groups):
            self.init_p2w[g, :3] = self.means[self.labels == g].mean(dim=0)
        self.part_deltas = torch.zeros(n_groups, 7, device=dev)
        self.part_deltas[:, 3] = 1.0
        self.part_deltas.requires_grad_(True)
        # Adam on the deltas (torch.optim.Adam defaults, rigid_group_optimizer.py:93) as one launch per iteration
        # together with the quaternion renormalisation (pr_pose_step); the state lives as long as the loop, as the
        # reference's part_optimizer does
        self.adam = {"step": 0, "exp_avg": torch.zeros_like(self.part_deltas), "betas": (0.9, 0.999), "eps": 1e-8,
                     "exp_avg_sq": torch.zeros_like(self.part_deltas)}
        self.target = None

    @classmethod
    def synthetic(cls, dev, n=200_000, width=640, height=360, groups=4, seed=0, offset=0.02, rois=False):
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
            if rois:   # per-object crops: the box of every group's projected Gaussians, inflated a little
                uv = sc.means @ sc.viewmats[0, :3, :3].T + sc.viewmats[0, :3, 3]
                uv = (uv[:, :2] / uv[:, 2:3]) * torch.stack([sc.Ks[0, 0, 0], sc.Ks[0, 1, 1]]) + sc.Ks[0, :2, 2]
                boxes = []
                for g_ in range(groups):
                    sel = uv[(labels == g_) & (uv[:, 0] > 0) & (uv[:, 0] < width) & (uv[:, 1] > 0) & (uv[:, 1] < height)]
                    lo, hi = sel.quantile(0.02, dim=0), sel.quantile(0.98, dim=0)
                    boxes.append((float((lo[0] / width).clamp(0, 0.9)), float((hi[0] / width).clamp(0.1, 1.0)),
                                  float((lo[1] / height).clamp(0, 0.9)), float((hi[1] / height).clamp(0.1, 1.0))))
                self.set_rois(boxes)
                obs = []
                for g_ in range(groups):
                    dino, depth, _ = self.render_roi(moved.to(dev), g_)
                    obs.append((dino[0], depth[0]))
                self.set_rois(boxes, obs)
        return self

    def render(self, deltas: Tensor, need_rgb: bool = True):
        """-> (features after the channel MLP [1,H,W,64], expected depth [1,H,W,1], rgb [1,H,W,3]; None when
        ``need_rgb`` is False and the renders share their binning: the first render is then "ED" instead of
        "RGB+ED")."""
        from .rendering import rasterization

        common = dict(viewmats=self.viewmat, Ks=self.K, width=self.width, height=self.height, packed=False)
        if self.fused_transform:
            m, q = self.means, self.quats
            common["rigid"] = (self.labels, self.init_p2w, deltas)
        else:
            m, q = apply_to_model(self.init_p2w, deltas, self.labels, self.means, self.quats)
        if self.fuse_renders:
            # Both renders differentiate through the means only (the reference's transform hands on quaternions without
            # gradient, rigid_group_optimizer.py:460-462, and the feature render detaches the rest, pogs/pogs.py:494-497)
            # and see the same camera: ONE render of [rgb | 64 features | depth] gives the same images and the same
            # pose gradients (up to what the tile size decides about the float32 tail of a list).
            out, alpha, _ = rasterization(m, q, self.scales, self.opacities, self.sh, sh_degree=3, render_mode="RGB+ED",
                                          tile_size=16, features=self.features, **common)
            F = self.features.shape[1]
            return self.net(out[..., 3:3 + F], alpha), out[..., 3 + F:], out[..., :3]
        if self.share_binning:
            # same positions, shapes, opacities, camera and image size: the feature render composites over the per-tile
            # lists of the RGB+ED render (both at tile 16, where the reference renders the first at 8: the image does
            # not depend on the tile size beyond the float32 tail of a list)
            # (without the colours: expected depth alone -- no view-dependent colours, one composited channel, the same
            # depth image)
            rgbd, _, meta = rasterization(m, q, self.scales, self.opacities, self.sh, sh_degree=3,
                                          render_mode="RGB+ED" if need_rgb else "ED", tile_size=self.shared_tile,
                                          **common)
            f, fa, _ = rasterization(m, q.detach(), self.scales, self.opacities, self.features,
                                     tile_size=self.shared_tile, binning=meta["binning"], **common)
        else:
            rgbd, _, meta = rasterization(m, q, self.scales, self.opacities, self.sh, sh_degree=3,
                                          render_mode="RGB+ED", tile_size=8, **common)
            # same positions, same camera: the feature render reuses the first render's depth order
            f, fa, _ = rasterization(m, q.detach(), self.scales, self.opacities, self.features,
                                     depth_order=meta["depth_order"], **common)
        if rgbd.shape[-1] == 1:
            return self.net(f, fa), rgbd, None
        return self.net(f, fa), rgbd[..., 3:4], rgbd[..., :3]

    def set_observation(self, dino: Tensor, depth: Optional[Tensor] = None) -> None:
        self.target = (dino.detach(), None if depth is None else depth.detach())

    # ---- per-object branch -------------------------------------------------------------------------------------
    @classmethod
    def crop_box(cls, box, width: int, height: int):
        """Normalised (xmin, xmax, ymin, ymax) -> pixel box with sides cut to multiples of the DINO patch, and the
        size it is rendered at (``observation.py:157-172``)."""
        xmin, xmax, ymin, ymax = box
        assert 0 <= xmin < xmax <= 1.0 and 0 <= ymin < ymax <= 1.0, "boxes are normalised to the image"
        x0, x1 = int(xmin * (width - 1)), int(xmax * (width - 1))
        y0, y1 = int(ymin * (height - 1)), int(ymax * (height - 1))
        w, h = (x1 - x0) // cls.DINO_PATCH * cls.DINO_PATCH, (y1 - y0) // cls.DINO_PATCH * cls.DINO_PATCH
        assert w > 0 and h > 0, "a crop must hold at least one 14-pixel patch per side"
        scale = min(1.0, cls.MAX_ROI_RESOLUTION / max(w, h))
        # (nerfstudio's rescale_output_resolution floors the new size)
        return x0, y0, w, h, scale, int(w * scale), int(h * scale)

    def set_rois(self, boxes, observations=None) -> None:
        """One normalised box per rigid group.  ``observations``: per group (dino [h,w,64], depth [h,w,1] or None) at
        the crop's render size; None keeps the crops without targets (``render_roi`` works, ``loss`` does not)."""
        assert len(boxes) == self.part_deltas.shape[0], "one box per rigid group"
        self.rois = []
        for g, box in enumerate(boxes):
            x0, y0, w, h, scale, rw, rh = self.crop_box(box, self.width, self.height)
            K = self.K.clone()
            K[:, 0, 2] -= x0          # crop_camera: the principal point moves with the corner
            K[:, 1, 2] -= y0
            K[:, :2, :] *= scale      # rescale_output_resolution
            ids = torch.where(self.labels == g)[0]
            self.rois.append(dict(K=K, width=rw, height=rh, ids=ids, box=(x0, y0, w, h),
                                  means=self.means[ids].contiguous(), quats=self.quats[ids].contiguous(),
                                  scales=self.scales[ids].contiguous(), opacities=self.opacities[ids].contiguous(),
                                  sh=self.sh[ids].contiguous(), features=self.features[ids].contiguous(),
                                  labels=self.labels[ids].contiguous(), target=None))
        self._roi_cams = self._roi_padded = None
        if observations is not None:
            for roi, (dino, depth) in zip(self.rois, observations):
                roi["target"] = (dino.detach(), None if depth is None else depth.detach())

    def render_roi(self, deltas: Tensor, g: int):
        """Group ``g`` alone through its crop camera -> (features after the channel MLP [1,h,w,64], expected depth
        [1,h,w,1], accumulation [1,h,w,1])."""
        from .rendering import rasterization

        roi = self.rois[g]
        common = dict(viewmats=self.viewmat, Ks=roi["K"], width=roi["width"], height=roi["height"], packed=False)
        if self.fused_transform:
            m, q = roi["means"], roi["quats"]
            common["rigid"] = (roi["labels"], self.init_p2w, deltas)
        else:
            m, q = apply_to_model(self.init_p2w, deltas, roi["labels"], roi["means"], roi["quats"])
        rgbd, acc, meta = rasterization(m, q, roi["scales"], roi["opacities"], roi["sh"], sh_degree=3,
                                        render_mode="RGB+ED", tile_size=8, **common)
        f, fa, _ = rasterization(m, q.detach(), roi["scales"], roi["opacities"], roi["features"],
                                 depth_order=meta["depth_order"], **common)
        return self.net(f, fa), rgbd[..., 3:4], acc

    @staticmethod
    def _erode(mask: Tensor, k: int) -> Tensor:
        """Binary erosion of an [h,w] mask by a k x k box; pixels beyond the border count as set (kornia's default)."""
        x = mask[None, None].float()
        return (-torch.nn.functional.max_pool2d(-x, k, stride=1, padding=k // 2))[0, 0] > 0.5

    def render_rois(self, deltas: Tensor):
        """All groups through their crop cameras in one call per pass: [G,Hm,Wm,*] images at the largest crop size
        (a crop's own pixels are its top-left [h,w] window; what lies beyond is rendered and ignored)."""
        from .rendering import rasterization

        G = len(self.rois)
        Wm, Hm = max(r["width"] for r in self.rois), max(r["height"] for r in self.rois)
        if getattr(self, "_roi_cams", None) is None:
            self._roi_cams = (self.viewmat.expand(G, 4, 4).contiguous(), torch.cat([r["K"] for r in self.rois]).contiguous(),
                              torch.arange(G, dtype=torch.int32, device=self.means.device))
        viewmats, Ks, groups = self._roi_cams
        common = dict(viewmats=viewmats, Ks=Ks, width=Wm, height=Hm, packed=False, camera_groups=groups,
                      rigid=(self.labels, self.init_p2w, deltas))
        rgbd, acc, meta = rasterization(self.means, self.quats, self.scales, self.opacities, self.sh, sh_degree=3,
                                        render_mode="RGB+ED", tile_size=16 if self.share_binning else 8, **common)
        shared = dict(binning=meta["binning"]) if self.share_binning else dict(depth_order=meta["depth_order"])
        f, fa, _ = rasterization(self.means, self.quats.detach(), self.scales, self.opacities, self.features,
                                 tile_size=16, **shared, **common)
        return self.net(f, fa), rgbd[..., 3:4], acc

    def _roi_targets(self):
        """Observations of all crops padded to the common size: features NaN beyond a crop (skipped by the mean),
        depths NaN beyond it (never valid), and the mask of each crop's own pixels."""
        if getattr(self, "_roi_padded", None) is None:
            G = len(self.rois)
            Wm, Hm = max(r["width"] for r in self.rois), max(r["height"] for r in self.rois)
            dev = self.means.device
            dino = torch.full((G, Hm, Wm, self.rois[0]["target"][0].shape[-1]), float("nan"), device=dev)
            depth = torch.full((G, Hm, Wm), float("nan"), device=dev)
            own = torch.zeros(G, Hm, Wm, dtype=torch.bool, device=dev)
            has_depth = all(r["target"][1] is not None for r in self.rois)
            for g, r in enumerate(self.rois):
                h, w = r["height"], r["width"]
                dino[g, :h, :w] = r["target"][0].reshape(h, w, -1)
                own[g, :h, :w] = True
                if has_depth:
                    depth[g, :h, :w] = r["target"][1].reshape(h, w)
            self._roi_padded = (dino, depth if has_depth else None, own)
        return self._roi_padded

    def _roi_loss_batched(self, use_depth: bool) -> Tensor:
        t_dino, t_depth, own = self._roi_targets()
        dino, depth, acc = self.render_rois(self.part_deltas)
        if not (use_depth and t_depth is not None):
            return tracking_loss(dino, t_dino)
        keep, masked = (roi_depth_masks if self.fused_roi_masks else roi_depth_masks_torch)(
            acc[..., 0].detach(), t_depth, self._roi_sizes(), own)
        return tracking_loss(dino, t_dino, depth[..., 0], masked, keep, self.dataset_scale, self.depth_ignore_threshold,
                             self.depth_loss_mult)

    def _roi_sizes(self) -> Tensor:
        if getattr(self, "_roi_size_table", None) is None:
            self._roi_size_table = torch.tensor([[r["height"], r["width"]] for r in self.rois], dtype=torch.int32,
                                                device=self.means.device)
        return self._roi_size_table

    def _roi_loss(self, use_depth: bool) -> Tensor:
        if self.batch_rois:
            return self._roi_loss_batched(use_depth)
        real_dino, rend_dino, real_depth, rend_depth, valids = [], [], [], [], []
        for g in reversed(range(len(self.rois))):
            roi = self.rois[g]
            assert roi["target"] is not None, "set_rois(..., observations=...) first"
            dino, depth, acc = self.render_roi(self.part_deltas, g)
            t_dino, t_depth = roi["target"]
            real_dino.append(t_dino.reshape(-1, t_dino.shape[-1]))
            rend_dino.append(dino.reshape(-1, dino.shape[-1]))
            if use_depth and t_depth is not None:
                obs = t_depth.reshape(roi["height"], roi["width"])
                valid = (acc.reshape(obs.shape) > 0.85) & ~obs.isnan()
                valid_depths = self._erode(valid, 5)
                picked = obs[valid_depths]
                if picked.numel() > 0:          # reject depths far behind the object's median
                    valid_depths = valid_depths & ~(obs > picked.median() * 1.3)
                masked = torch.where(valid_depths, obs, torch.zeros_like(obs))
                nonzero = masked != 0
                keep = self._erode(valid_depths & nonzero, 9)
                real_depth.append(torch.where(keep, masked, torch.zeros_like(masked)).reshape(-1))
                rend_depth.append(depth.reshape(-1))
                valids.append(keep.reshape(-1))
        if valids:
            return tracking_loss(torch.cat(rend_dino), torch.cat(real_dino), torch.cat(rend_depth), torch.cat(real_depth),
                                 torch.cat(valids), self.dataset_scale, self.depth_ignore_threshold, self.depth_loss_mult)
        return tracking_loss(torch.cat(rend_dino), torch.cat(real_dino))

    def loss(self, use_depth: bool = True) -> Tensor:
        if self.rois is not None:
            return self._roi_loss(use_depth)
        # the reference renders the colours whether or not a loss reads them (get_outputs always does); skip_unused_rgb
        # leaves them out of the optimisation's renders: same loss, same gradients
        dino, depth, _ = self.render(self.part_deltas, need_rgb=not self.skip_unused_rgb)
        if use_depth and self.target[1] is not None:
            return tracking_loss(dino, self.target[0], depth, self.target[1], None, self.dataset_scale,
                                 self.depth_ignore_threshold, self.depth_loss_mult)
        return tracking_loss(dino, self.target[0])

    def frame(self, niter: int = 50, use_depth: bool = True):
        """One frame's ``niter`` iterations; returns the loss of every iteration (read back once, at the end)."""
        assert self.target is not None, "set_observation first"
        gamma = (self.pose_lr_final / self.pose_lr) ** (1.0 / max(niter, 1))
        losses = []
        with torch.no_grad():       # "renormalize rotation representation" (:411-412) of the first iteration; the
            # later ones get theirs from the step before them
            self.part_deltas[:, 3:] = self.part_deltas[:, 3:] / self.part_deltas[:, 3:].norm(dim=1, keepdim=True)
        for i in range(niter):
            self.part_deltas.grad = None
            loss = self.loss(use_depth)
            loss.backward()
            # the scheduler's rate of iteration i (ExponentialDecayScheduler, :403-407); the rate is back at pose_lr
            # when the next frame starts (:441)
            self.pose_step(self.pose_lr * gamma ** i, renormalise=i + 1 < niter)
            losses.append(loss.detach())
        return torch.stack(losses).tolist()

    @torch.no_grad()
    def pose_step(self, lr: float, renormalise: bool) -> None:
        """``part_optimizer.step()`` (:432) and, with ``renormalise``, the renormalisation that opens the next
        iteration, as one launch."""
        from .rendering import _call, _ptr

        st, g = self.adam, self.part_deltas.grad
        if g is None:
            return                  # torch.optim.Adam skips a parameter without gradient
        g = g.contiguous()
        st["step"] += 1
        _call("pr_pose_step", self.part_deltas.shape[0], _ptr(self.part_deltas), _ptr(g), _ptr(st["exp_avg"]),
              _ptr(st["exp_avg_sq"]), st["step"], float(lr), st["betas"][0], st["betas"][1], st["eps"],
              1 if renormalise else 0, torch.cuda.current_stream(g.device).cuda_stream)
