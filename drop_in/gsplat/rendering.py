"""``gsplat.rendering`` as imported by ``pogs/pogs.py:11`` and nerfstudio's ``splatfacto.py:29``."""
from pogs_amd.rendering import (  # noqa: F401
    fully_fused_projection, isect_offset_encode, isect_tiles, rasterization, rasterize_to_pixels)
