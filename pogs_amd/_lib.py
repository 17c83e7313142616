"""ctypes binding of ``libpogs_rast.so`` (the C ABI declared in ``include/pogs_rast.h``).

The library is built in-tree by ``pogs_amd/csrc/Makefile`` (``__graft_entry__.build()`` drives it).
There is NO fallback: if the shared object is missing or a symbol is absent, importing the
rasterizer fails loudly.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# POGS_RAST_LIB names another build of the same library (e.g. the cycle-stamp diagnostic build, EXTRA=-DPR_DIAG=2)
LIB_PATH = os.environ.get("POGS_RAST_LIB") or os.path.join(_HERE, "lib", "libpogs_rast.so")

_P = c_void_p

# name -> (restype, argtypes); mirrors include/pogs_rast.h one to one.
SIGNATURES = {
    "pr_version": (c_int, []),
    "pr_error_string": (c_char_p, [c_int]),
    "pr_last_hip_error": (c_char_p, []),
    "pr_max_channels": (c_int, []),
    "pr_reload_tuning": (None, []),
    "pr_project_fwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float, c_float, c_float,
                               _P, _P, _P, _P, _P, _P]),
    "pr_project_bwd": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P, _P, c_int, _P, _P, c_int,
                               _P, _P, _P, _P, _P, _P]),
    "pr_sh_fwd": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, c_int, _P]),
    "pr_sh_bwd": (c_int, [c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, c_int, _P, _P, _P, _P]),
    "pr_view_depths": (c_int, [c_int, c_int, _P, _P, _P, _P]),
    "pr_campos_fwd": (c_int, [c_int, _P, _P, _P]),
    "pr_campos_bwd": (c_int, [c_int, _P, _P, _P, _P]),
    "pr_isect_count_scratch_bytes": (c_size_t, [c_int, c_int]),
    "pr_isect_count": (c_int, [c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "pr_isect_sort_scratch_bytes": (c_size_t, [c_int, c_int, c_int64, c_int64]),
    "pr_isect_emit_sort": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int64, _P, _P, _P,
                                   _P, _P, _P, _P, c_size_t, _P]),
    "pr_tile_lists_count": (c_int, [c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "pr_tile_lists_fill": (c_int, [c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int64, _P, _P, _P,
                                   _P]),
    "pr_publish_count": (c_int, [_P, _P, c_int64, _P]),
    "pr_depth_order_scratch_bytes": (c_size_t, [c_int, c_int]),
    "pr_depth_order": (c_int, [c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "pr_depth_order_static_max": (c_int64, []),
    "pr_depth_order_static": (c_int, [c_int, c_int, _P, _P, _P]),
    "pr_isect_offsets": (c_int, [c_int64, _P, c_int, c_int, c_int, _P, _P]),
    "pr_raster_fwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P,
                              c_int64, c_int, _P, _P, _P, _P, c_int, _P]),
    "pr_raster_bwd": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P,
                              c_int64, _P, _P, _P, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P, _P, c_int, _P]),
    "pr_raster_fwd_plan": (c_int, [c_int, c_int, c_int64, _P, _P, _P]),
    "pr_raster_bwd_plan": (c_int, [c_int, c_int, c_int64, c_int, c_int, _P, _P, _P]),
    "pr_mlp_mask_words": (c_size_t, [c_int64]),
    "pr_mlp_fwd": (c_int, [c_int64, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "pr_mlp_bwd_input": (c_int, [c_int64, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "pr_mlp_fwd_train": (c_int, [c_int64, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "pr_mlp_bwd_full": (c_int, [c_int64, _P, _P, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "pr_dense_packed_bytes": (c_size_t, [c_int, c_int]),
    "pr_dense_pack": (c_int, [c_int, c_int, _P, c_int64, c_int64, _P, _P]),
    "pr_dense_fwd": (c_int, [c_int64, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    "pr_dense_wgrad": (c_int, [c_int64, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P]),
    "pr_l2norm_fwd": (c_int, [c_int64, c_int, _P, c_float, _P, _P, _P]),
    "pr_l2norm_bwd": (c_int, [c_int64, c_int, _P, _P, c_float, _P, _P, _P]),
    "pr_densify_accumulate": (c_int, [c_int, c_int, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pr_densify_plan_scratch_bytes": (c_size_t, [c_int]),
    "pr_densify_plan": (c_int, [c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "pr_densify_emit": (c_int, [c_int, _P, c_size_t, c_int64, _P, _P, _P]),
    "pr_densify_gather": (c_int, [c_int64, c_int, _P, _P, _P, c_int, _P, _P]),
    "pr_densify_split_children": (c_int, [c_int64, _P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pr_adam_max_groups": (c_int, []),
    "pr_adam_step": (c_int, [_P, c_int, _P]),
    "pr_project_fwd_rigid": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, c_float,
                                     c_float, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "pr_project_bwd_rigid": (c_int, [c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P, _P,
                                     c_int, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "pr_track_loss_fwd": (c_int, [c_int64, c_int, _P, _P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P]),
    "pr_track_loss_bwd": (c_int, [c_int64, c_int, _P, _P, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P, _P, _P]),
    "pr_rigid_fwd": (c_int, [c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pr_rigid_bwd": (c_int, [c_int64, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pr_roi_depth_masks_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pr_roi_depth_masks": (c_int, [c_int, c_int, c_int, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P]),
    "pr_pose_step": (c_int, [c_int, _P, _P, _P, _P, c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                             ctypes.c_double, c_int, _P]),
}



class Segment(ctypes.Structure):
    """pr_color_segment / pr_grad_segment (same layout): one of up to three tensors that together supply the
    composited channels."""
    _fields_ = [("ptr", ctypes.c_void_p), ("width", c_int), ("ld", c_int), ("shared", c_int)]


class AdamGroup(ctypes.Structure):
    """pr_adam_group."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("numel", c_int64), ("step", c_int64), ("lr", ctypes.c_double),
                ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double)]


class DensifyRule(ctypes.Structure):
    """pr_densify_rule."""
    _fields_ = [("grow_grad2d", c_float), ("grow_scale3d", c_float), ("grow_scale2d", c_float),
                ("prune_opa", c_float), ("prune_scale3d", c_float), ("prune_scale2d", c_float),
                ("grow_by_radius", c_int), ("prune_big", c_int), ("prune_by_radius", c_int),
                ("revised_opacity", c_int)]


def segment_array(items):
    """[(data_ptr or None, width, ld, shared)] -> (ctypes array kept alive by the caller, void pointer)."""
    arr = (Segment * len(items))(*[Segment(p, w, ld, int(sh)) for p, w, ld, sh in items])
    return arr, ctypes.cast(arr, ctypes.c_void_p)


_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library once and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C pogs_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "pogs_amd has no CPU or PyTorch fallback for the rasterizer.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        lib = load()
        msg = lib.pr_error_string(rc).decode()
        detail = lib.pr_last_hip_error().decode()
        if rc == -4 or (rc == -2 and detail.startswith(what)):
            msg += ": " + detail
        raise RuntimeError(f"{what} failed ({rc}): {msg}")


def raster_plan(D: int, tile_size: int, n_tiles_total: int, backward: bool = False, geom_grad: bool = True,
                color_grad: bool = True):
    """(pixels per lane, channels per pass, matrix cores?) of the compositor kernel the library picks."""
    ppl, per, mm = c_int(0), c_int(0), c_int(0)
    lib = load()
    if backward:
        rc = lib.pr_raster_bwd_plan(D, tile_size, n_tiles_total, int(geom_grad), int(color_grad),
                                    ctypes.addressof(ppl), ctypes.addressof(per), ctypes.addressof(mm))
    else:
        rc = lib.pr_raster_fwd_plan(D, tile_size, n_tiles_total, ctypes.addressof(ppl), ctypes.addressof(per),
                                    ctypes.addressof(mm))
    check(rc, "pr_raster_plan")
    return ppl.value, per.value, bool(mm.value)


def set_tuning(**overrides) -> None:
    """Set kernel-variant overrides for this process: ``set_tuning(POGS_RAST_BWD_PPL=4)`` puts the value in the
    environment (``None`` removes it) and has the library re-read its tuning variables, which it otherwise reads
    only once.  For tests and A/B scripts; the render path never calls this."""
    for k, v in overrides.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    load().pr_reload_tuning()
