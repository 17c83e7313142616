"""CPU oracle for the rigid-group transform.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

PARITY UNPINNED.  The reference computes this in an NVIDIA Warp kernel, ``apply_to_model``
(``pogs/tracking/utils.py:53-93``), and differentiates it with ``wp.Tape``
(``pogs/tracking/rigid_group_optimizer.py:413-430``).  Warp (``warp-lang``, unpinned in the reference's
requirements) is not installed here and the reference holds no fixture for this kernel, so the functions
below restate Warp's published transform algebra and are pinned only by closed-form known answers and by
agreement between two formulations (quaternion algebra, homogeneous matrices):

* ``quat_rotate(q, v) = v (2 w^2 - 1) + 2 w (q_v x v) + 2 q_v (q_v . v)``  (no renormalisation of ``q``),
* ``quat_inverse`` = conjugate, Hamilton product with the vector part stored first,
* ``transform_multiply(a, b) = (quat_rotate(a.q, b.p) + a.p, a.q * b.q)``,
* ``transform_inverse(t) = (-quat_rotate(conj(t.q), t.p), conj(t.q))``.

Gradients come from autograd on these formulas in float64.
"""
from __future__ import annotations

import torch
from torch import Tensor


def _cross(a: Tensor, b: Tensor) -> Tensor:
    return torch.cross(a, b, dim=-1)


def quat_rotate(q_xyzw: Tensor, v: Tensor) -> Tensor:
    qv, w = q_xyzw[..., :3], q_xyzw[..., 3:]
    return v * (2.0 * w * w - 1.0) + _cross(qv, v) * w * 2.0 + qv * (qv * v).sum(-1, keepdim=True) * 2.0


def quat_mul(a: Tensor, b: Tensor) -> Tensor:
    av, aw, bv, bw = a[..., :3], a[..., 3:], b[..., :3], b[..., 3:]
    return torch.cat([aw * bv + bw * av + _cross(av, bv), aw * bw - (av * bv).sum(-1, keepdim=True)], dim=-1)


def quat_conj(q: Tensor) -> Tensor:
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def transform_multiply(a, b):
    return quat_rotate(a[1], b[0]) + a[0], quat_mul(a[1], b[1])


def transform_inverse(t):
    qi = quat_conj(t[1])
    return -quat_rotate(qi, t[0]), qi


def _pose(poses: Tensor, idx: Tensor):
    """(x y z qw qx qy qz) rows -> (position, quaternion xyzw), as ``poses_7vec_to_transform``
    (pogs/tracking/utils.py:44-51)."""
    rows = poses[idx]
    return rows[:, :3], torch.cat([rows[:, 4:7], rows[:, 3:4]], dim=-1)


def apply_to_model(init_p2ws: Tensor, p_deltas: Tensor, group_labels: Tensor, means: Tensor, quats: Tensor):
    """Restates pogs/tracking/utils.py:79-93 for all Gaussians at once.  quats and the result are (w x y z)."""
    idx = group_labels.long()
    p2w, delta = _pose(init_p2ws, idx), _pose(p_deltas, idx)
    g2w = (means, torch.cat([quats[:, 1:4], quats[:, 0:1]], dim=-1))
    g2p = transform_multiply(transform_inverse(p2w), g2w)
    new = transform_multiply(transform_multiply(p2w, delta), g2p)
    return new[0], torch.cat([new[1][:, 3:4], new[1][:, :3]], dim=-1)


def _rotmat(q_wxyz: Tensor) -> Tensor:
    w, x, y, z = q_wxyz.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def _homogeneous(p: Tensor, q_wxyz: Tensor) -> Tensor:
    m = torch.zeros(p.shape[0], 4, 4, dtype=p.dtype)
    m[:, :3, :3] = _rotmat(q_wxyz)
    m[:, :3, 3] = p
    m[:, 3, 3] = 1
    return m


def apply_to_model_matrices(init_p2ws: Tensor, p_deltas: Tensor, group_labels: Tensor, means: Tensor,
                            quats: Tensor):
    """Second formulation, valid for UNIT quaternions: 4x4 matrices, new = P D P^-1 G.  Returns the new means
    and the new rotation matrices [N,3,3]."""
    idx = group_labels.long()
    P = _homogeneous(init_p2ws[idx, :3], init_p2ws[idx, 3:7])
    D = _homogeneous(p_deltas[idx, :3], p_deltas[idx, 3:7])
    Gm = _homogeneous(means, quats)
    new = P @ D @ torch.linalg.inv(P) @ Gm
    return new[:, :3, 3], new[:, :3, :3]
