"""``gsplat.cuda._torch_impl._quat_to_rotmat`` -- the one helper the reference's GUI code imports
(``pogs/pogs_pipeline.py:29``).  Plain torch, device agnostic."""
import torch
import torch.nn.functional as F


def _quat_to_rotmat(quats: torch.Tensor) -> torch.Tensor:
    """wxyz quaternions [..., 4] (normalised inside) -> rotation matrices [..., 3, 3]; the same matrix as
    ``pogs/tracking/utils.py:101-121``."""
    w, x, y, z = torch.unbind(F.normalize(quats, p=2, dim=-1), dim=-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(quats.shape[:-1] + (3, 3))
