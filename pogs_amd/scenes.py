"""Deterministic synthetic scenes shaped like the POGS workloads, and the host glue that sits
either side of the rasterizer call in the reference model.

Nothing here touches the GPU library; tensors are generated on the CPU with a seeded generator
(so CPU oracle and GPU path see bit-identical inputs) and moved by the caller.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

SH_C0 = 0.28209479177387814


def rgb2sh(rgb: Tensor) -> Tensor:
    """Same mapping as nerfstudio's ``RGB2SH``
    (``src/pogs/dependencies/nerfstudio/nerfstudio/utils/spherical_harmonics.py:98-104``)."""
    return (rgb - 0.5) / SH_C0


def get_viewmat(camera_to_world: Tensor) -> Tensor:
    """nerfstudio camera-to-world (OpenGL axes) -> world-to-camera (OpenCV axes) 4x4, the same
    convention as ``get_viewmat`` in ``pogs/pogs.py:75-90``: negate the y and z columns of R,
    then invert the rigid transform analytically.  camera_to_world: [C,3,4] or [C,4,4]."""
    R = camera_to_world[:, :3, :3]
    T = camera_to_world[:, :3, 3:4]
    flip = torch.tensor([[[1.0, -1.0, -1.0]]], dtype=R.dtype, device=R.device)
    R = R * flip
    R_inv = R.transpose(1, 2)
    T_inv = -torch.bmm(R_inv, T)
    viewmat = torch.zeros(R.shape[0], 4, 4, dtype=R.dtype, device=R.device)
    viewmat[:, 3, 3] = 1.0
    viewmat[:, :3, :3] = R_inv
    viewmat[:, :3, 3:4] = T_inv
    return viewmat


def composite_background(render: Tensor, alpha: Tensor, background: Tensor) -> Tensor:
    """``rgb = clamp(render[..., :3] + (1 - alpha) * background, 0, 1)`` as in ``pogs/pogs.py:382-383``."""
    return torch.clamp(render[..., :3] + (1 - alpha) * background, 0.0, 1.0)


def fill_depth(render: Tensor, alpha: Tensor) -> Tensor:
    """Expected-depth channel with empty pixels set to the image maximum (``pogs/pogs.py:386-387``)."""
    d = render[..., 3:4]
    return torch.where(alpha > 0, d, d.detach().max())


def normalise_features(feats: Tensor, alpha: Tensor) -> Tensor:
    """``feat / clamp(alpha, 1e-6)`` where alpha > 0 else 0 (``pogs/pogs.py:516-518``)."""
    a = torch.clamp(alpha, min=1e-6)
    return torch.where(alpha > 0, feats / a.detach(), torch.zeros_like(feats))


@dataclass
class Scene:
    means: Tensor        # [N,3]
    quats: Tensor        # [N,4] wxyz, unnormalised
    log_scales: Tensor   # [N,3]  (call site applies exp)
    opacity_logits: Tensor  # [N] (call site applies sigmoid)
    sh: Tensor           # [N,16,3]
    features: Tensor     # [N,F]
    viewmats: Tensor     # [C,4,4]
    Ks: Tensor           # [C,3,3]
    width: int
    height: int

    @property
    def n(self) -> int:
        return self.means.shape[0]

    def to(self, device) -> "Scene":
        return Scene(*[t.to(device) if isinstance(t, Tensor) else t for t in
                       (self.means, self.quats, self.log_scales, self.opacity_logits, self.sh,
                        self.features, self.viewmats, self.Ks)], self.width, self.height)


def ring_viewmats(n_views: int, radius: float = 3.0, dtype=torch.float32) -> Tensor:
    """World-to-camera matrices for cameras on a ring of the given radius in the xz-plane, all
    looking at the origin (OpenCV axes: +z forward, +y down).  View 0 sits at (0,0,-radius) with
    identity rotation, i.e. viewmat = [I | (0,0,radius)]."""
    mats = []
    for i in range(n_views):
        th = 2.0 * math.pi * i / max(n_views, 1)
        # camera position
        pos = torch.tensor([-radius * math.sin(th), 0.0, -radius * math.cos(th)], dtype=torch.float64)
        fwd = -pos / pos.norm()
        up = torch.tensor([0.0, -1.0, 0.0], dtype=torch.float64)  # -y is up (y points down in images)
        right = torch.linalg.cross(fwd, up)
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        R_wc = torch.stack([right, down, fwd], dim=0)  # rows = camera axes in world coords
        t = -R_wc @ pos
        m = torch.eye(4, dtype=torch.float64)
        m[:3, :3] = R_wc
        m[:3, 3] = t
        mats.append(m)
    return torch.stack(mats, 0).to(dtype)


def make_scene(n: int, width: int, height: int, n_features: int = 32, n_views: int = 1, seed: int = 0,
               scale_lo: Optional[float] = None, scale_hi: Optional[float] = None) -> Scene:
    """Random scene: means ~ U([-1,1]^3) seen from distance 3; random unnormalised quats;
    log-scales ~ U(log lo, log hi) with lo/hi = 0.002/0.02 at N = 1e6 scaled by (1e6/N)^(1/3) so the
    mean number of tiles per Gaussian stays roughly constant; opacity logits ~ N(0,1.5);
    SH dc from U(0,1) colours, higher bands ~ N(0,0.1); features ~ N(0,1);
    pinhole fx = fy = 0.9 W, principal point at the image centre."""
    g = torch.Generator().manual_seed(seed)
    f = (1.0e6 / max(n, 1)) ** (1.0 / 3.0)
    lo = 0.002 * f if scale_lo is None else scale_lo
    hi = 0.02 * f if scale_hi is None else scale_hi
    means = torch.rand(n, 3, generator=g) * 2.0 - 1.0
    quats = torch.randn(n, 4, generator=g)
    log_scales = torch.rand(n, 3, generator=g) * (math.log(hi) - math.log(lo)) + math.log(lo)
    opacity_logits = torch.randn(n, generator=g) * 1.5
    dc = rgb2sh(torch.rand(n, 1, 3, generator=g))
    rest = torch.randn(n, 15, 3, generator=g) * 0.1
    sh = torch.cat([dc, rest], dim=1)
    features = torch.randn(n, n_features, generator=g)
    viewmats = ring_viewmats(n_views)
    K = torch.tensor([[0.9 * width, 0.0, width / 2.0], [0.0, 0.9 * width, height / 2.0], [0.0, 0.0, 1.0]])
    Ks = K[None].repeat(n_views, 1, 1)
    return Scene(means, quats, log_scales, opacity_logits, sh, features, viewmats, Ks, width, height)
