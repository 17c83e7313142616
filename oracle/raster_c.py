"""ctypes wrapper of ``oracle/raster_c.c`` (CPU float32 scalar restatement of the tile compositor).

TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see ``oracle/__init__.py``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

import torch
from torch import Tensor

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liboracle_raster.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", _HERE], check=True)
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def raster_fwd(means2d, conics, colors, opacities, width, height, ts, isect_offsets, flatten_ids,
               backgrounds=None, tiles: Optional[Tensor] = None):
    """Same contract as ``gsplat_ref.rasterize_to_pixels`` (float32 only) plus last_ids; ``tiles`` restricts
    the work to the listed flat tile indices (other pixels are left zero)."""
    lib = load()
    C, N, D = colors.shape
    th, tw = isect_offsets.shape[1:]
    m, cn, cl, op = _f32(means2d), _f32(conics), _f32(colors), _f32(opacities)
    bg = None if backgrounds is None else _f32(backgrounds)
    offs = isect_offsets.to(torch.int32).contiguous()
    flat = flatten_ids.to(torch.int32).contiguous()
    til = None if tiles is None else tiles.to(torch.int32).contiguous()
    out = torch.zeros(C, height, width, D)
    alpha = torch.zeros(C, height, width, 1)
    last = torch.zeros(C, height, width, dtype=torch.int32)
    lib.oracle_raster_fwd(C, N, D, _p(m), _p(cn), _p(cl), _p(op), _p(bg), width, height, ts, tw, th, _p(offs),
                          _p(flat), ctypes.c_int64(flat.numel()), _p(til), 0 if til is None else til.numel(),
                          _p(out), _p(alpha), _p(last))
    return out, alpha, last


def raster_bwd(means2d, conics, colors, opacities, width, height, ts, isect_offsets, flatten_ids, render_alphas,
               last_ids, v_render_colors, v_render_alphas, backgrounds=None, tiles: Optional[Tensor] = None):
    """-> v_means2d [C,N,2], v_means2d_abs [C,N,2], v_conics [C,N,3], v_colors [C,N,D], v_opacities [C,N]."""
    lib = load()
    C, N, D = colors.shape
    th, tw = isect_offsets.shape[1:]
    m, cn, cl, op = _f32(means2d), _f32(conics), _f32(colors), _f32(opacities)
    bg = None if backgrounds is None else _f32(backgrounds)
    offs = isect_offsets.to(torch.int32).contiguous()
    flat = flatten_ids.to(torch.int32).contiguous()
    til = None if tiles is None else tiles.to(torch.int32).contiguous()
    ra, li = _f32(render_alphas), last_ids.to(torch.int32).contiguous()
    vc, va = _f32(v_render_colors), _f32(v_render_alphas)
    v_m = torch.zeros(C, N, 2)
    v_ma = torch.zeros(C, N, 2)
    v_cn = torch.zeros(C, N, 3)
    v_cl = torch.zeros(C, N, D)
    v_op = torch.zeros(C, N)
    lib.oracle_raster_bwd(C, N, D, _p(m), _p(cn), _p(cl), _p(op), _p(bg), width, height, ts, tw, th, _p(offs),
                          _p(flat), ctypes.c_int64(flat.numel()), _p(til), 0 if til is None else til.numel(),
                          _p(ra), _p(li), _p(vc), _p(va), _p(v_m), _p(v_ma), _p(v_cn), _p(v_cl), _p(v_op))
    return v_m, v_ma, v_cn, v_cl, v_op
