"""Pure-PyTorch CPU restatement of gsplat 1.4.0's rasterization path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): PARITY UNPINNED, the
reference holds neither gsplat's source nor golden vectors for this path.

Every function names the upstream gsplat 1.4.0 routine it restates
(``gsplat/cuda/_torch_impl.py`` unless noted) and the reference call site
that exercises it.  Everything is dtype-generic: run in float64 with autograd
to get golden gradients, in float32 to mimic the device arithmetic.

Reference call sites: ``pogs/pogs.py:352-371`` (RGB / RGB+ED, SH),
``pogs/pogs.py:419-437`` (96-ch features, absgrad), ``pogs/pogs.py:493-510``
(64-ch features), ``pogs/pogs.py:930-948`` (no-grad 96-ch),
``pogs/tracking/rigid_group_optimizer.py:299,314,545`` (tile_size 8).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

ALPHA_MIN = 1.0 / 255.0
ALPHA_MAX = 0.999
T_STOP = 1e-4
AMBIG_TOL = 2e-4  # relative half-width of the "either branch is acceptable" band around a threshold


# --------------------------------------------------------------------------
# A.2 projection
# --------------------------------------------------------------------------
def quat_to_rotmat(quats: Tensor) -> Tensor:
    """gsplat ``_quat_to_rotmat`` (wxyz, normalised inside).  Same matrix as
    ``pogs/tracking/utils.py:101-121``."""
    q = F.normalize(quats, p=2, dim=-1)
    w, x, y, z = torch.unbind(q, dim=-1)
    R = torch.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y),
        ],
        dim=-1,
    )
    return R.reshape(quats.shape[:-1] + (3, 3))


def quat_scale_to_covar(quats: Tensor, scales: Tensor) -> Tensor:
    """gsplat ``_quat_scale_to_covar_preci`` (covariance only): (R S)(R S)^T."""
    R = quat_to_rotmat(quats)
    M = R * scales[..., None, :]
    return M @ M.transpose(-1, -2)


def world_to_cam(means: Tensor, covars: Tensor, viewmats: Tensor) -> Tuple[Tensor, Tensor]:
    """gsplat ``_world_to_cam``: p_c = R p + t, S_c = R S R^T.  -> [C,N,3], [C,N,3,3]."""
    R = viewmats[:, :3, :3]
    t = viewmats[:, :3, 3]
    means_c = torch.einsum("cij,nj->cni", R, means) + t[:, None, :]
    covars_c = torch.einsum("cij,njk,clk->cnil", R, covars, R)
    return means_c, covars_c


def persp_proj(means_c: Tensor, covars_c: Tensor, Ks: Tensor, width: int, height: int):
    """gsplat ``_persp_proj``: EWA pinhole projection with the 0.3*tanfov clamp
    applied only inside the Jacobian."""
    tx, ty, tz = torch.unbind(means_c, dim=-1)
    tz2 = tz * tz
    fx = Ks[:, 0, 0, None]
    fy = Ks[:, 1, 1, None]
    cx = Ks[:, 0, 2, None]
    cy = Ks[:, 1, 2, None]
    tan_fovx = 0.5 * width / fx
    tan_fovy = 0.5 * height / fy
    lim_x_pos = (width - cx) / fx + 0.3 * tan_fovx
    lim_x_neg = cx / fx + 0.3 * tan_fovx
    lim_y_pos = (height - cy) / fy + 0.3 * tan_fovy
    lim_y_neg = cy / fy + 0.3 * tan_fovy
    txc = tz * torch.maximum(torch.minimum(tx / tz, lim_x_pos), -lim_x_neg)
    tyc = tz * torch.maximum(torch.minimum(ty / tz, lim_y_pos), -lim_y_neg)
    O = torch.zeros_like(tz)
    J = torch.stack(
        [fx / tz, O, -fx * txc / tz2, O, fy / tz, -fy * tyc / tz2], dim=-1
    ).reshape(tz.shape + (2, 3))
    cov2d = J @ covars_c @ J.transpose(-1, -2)
    means2d = torch.stack([fx * tx / tz + cx, fy * ty / tz + cy], dim=-1)
    return means2d, cov2d


def fully_fused_projection(
    means: Tensor, quats: Tensor, scales: Tensor, viewmats: Tensor, Ks: Tensor,
    width: int, height: int, eps2d: float = 0.3, near_plane: float = 0.01,
    far_plane: float = 1e10, radius_clip: float = 0.0, calc_compensations: bool = False,
    radii_override: Optional[Tensor] = None,
):
    """gsplat ``_fully_fused_projection`` (non-packed).  ``radii_override`` (test aid) replaces the
    computed integer radii so that a comparison against a float32 implementation shares its discrete
    culling/binning decisions.  Returns
    radii[C,N] int32, means2d[C,N,2], depths[C,N], conics[C,N,3], compensations|None."""
    covars = quat_scale_to_covar(quats, scales)
    means_c, covars_c = world_to_cam(means, covars, viewmats)
    means2d, cov2d = persp_proj(means_c, covars_c, Ks, width, height)
    det_orig = cov2d[..., 0, 0] * cov2d[..., 1, 1] - cov2d[..., 0, 1] * cov2d[..., 1, 0]
    cov2d = cov2d + torch.eye(2, dtype=cov2d.dtype) * eps2d
    det = cov2d[..., 0, 0] * cov2d[..., 1, 1] - cov2d[..., 0, 1] * cov2d[..., 1, 0]
    det = det.clamp(min=1e-10)
    compensations = None
    if calc_compensations:
        compensations = torch.sqrt(torch.clamp(det_orig / det, min=0.0))
    conics = torch.stack(
        [cov2d[..., 1, 1] / det, -(cov2d[..., 0, 1] + cov2d[..., 1, 0]) / 2.0 / det, cov2d[..., 0, 0] / det],
        dim=-1,
    )
    depths = means_c[..., 2]
    b = (cov2d[..., 0, 0] + cov2d[..., 1, 1]) / 2
    v1 = b + torch.sqrt(torch.clamp(b * b - det, min=0.01))
    radius = torch.ceil(3.0 * torch.sqrt(v1)).detach()
    valid = (det > 0) & (depths > near_plane) & (depths < far_plane)
    radius = torch.where(valid, radius, torch.zeros_like(radius))
    m = means2d.detach()
    inside = (
        (m[..., 0] + radius > 0) & (m[..., 0] - radius < width)
        & (m[..., 1] + radius > 0) & (m[..., 1] - radius < height)
    )
    radius = torch.where(inside, radius, torch.zeros_like(radius))
    radius = torch.where(radius <= radius_clip, torch.zeros_like(radius), radius)
    radius = torch.nan_to_num(radius, nan=0.0, posinf=0.0, neginf=0.0)
    radii = radius.to(torch.int32)
    if radii_override is not None:
        radii = radii_override.to(torch.int32)
    return radii, means2d, depths, conics, compensations


# --------------------------------------------------------------------------
# A.3 spherical harmonics (Sloan 2013 fast basis, degree <= 3)
# --------------------------------------------------------------------------
def eval_sh_bases(num_bases: int, dirs: Tensor) -> Tensor:
    """gsplat ``_eval_sh_bases_fast``; dirs must be unit vectors.  -> [..., num_bases].
    Magnitudes agree with nerfstudio's own basis
    (``src/pogs/dependencies/nerfstudio/nerfstudio/utils/spherical_harmonics.py:49-73``)."""
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    out = [torch.full_like(x, 0.2820947917738781)]
    if num_bases > 1:
        fTmpA = -0.48860251190292
        out += [fTmpA * y, -fTmpA * z, fTmpA * x]
    if num_bases > 4:
        z2 = z * z
        fTmpB = -1.092548430592079 * z
        fTmpA2 = 0.5462742152960395
        fC1 = x * x - y * y
        fS1 = 2 * x * y
        out += [fTmpA2 * fS1, fTmpB * y, 0.9461746957575601 * z2 - 0.3153915652525201, fTmpB * x, fTmpA2 * fC1]
        if num_bases > 9:
            fTmpC = -2.285228997322329 * z2 + 0.4570457994644658
            fTmpB3 = 1.445305721320277 * z
            fTmpA3 = -0.5900435899266435
            fC2 = x * fC1 - y * fS1
            fS2 = x * fS1 + y * fC1
            out += [
                fTmpA3 * fS2, fTmpB3 * fS1, fTmpC * y,
                z * (1.865881662950577 * z2 - 1.119528997770346),
                fTmpC * x, fTmpB3 * fC1, fTmpA3 * fC2,
            ]
            if num_bases > 16:
                raise NotImplementedError("SH degree 4 is outside the path (POGS caps at 3)")
    return torch.stack(out, dim=-1)


def spherical_harmonics(degree: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None) -> Tensor:
    """gsplat ``_spherical_harmonics``: dirs [...,3] (unnormalised), coeffs [...,K,3] -> [...,3].
    Masked-out entries produce 0."""
    nb = (degree + 1) ** 2
    assert nb <= coeffs.shape[-2]
    d = F.normalize(dirs, p=2, dim=-1)
    bases = eval_sh_bases(nb, d)
    col = (bases[..., None] * coeffs[..., :nb, :]).sum(dim=-2)
    if masks is not None:
        col = torch.where(masks[..., None], col, torch.zeros_like(col))
    return col


# --------------------------------------------------------------------------
# A.4 / A.5 tile binning
# --------------------------------------------------------------------------
def tile_rects(means2d: Tensor, radii: Tensor, ts: int, tw: int, th: int):
    """Tile bounding rectangles [tmin, tmax) as in gsplat ``_isect_tiles``; float32 arithmetic
    regardless of the input dtype, so float64 oracle runs bin exactly like float32 ones."""
    m = means2d.detach().to(torch.float32)
    r = radii.to(torch.float32)
    tile_means = m / float(ts)
    tile_r = (r / float(ts))[..., None]
    tmin = torch.floor(tile_means - tile_r).to(torch.int64)
    tmax = torch.ceil(tile_means + tile_r).to(torch.int64)
    tmin_x = tmin[..., 0].clamp(0, tw)
    tmin_y = tmin[..., 1].clamp(0, th)
    tmax_x = tmax[..., 0].clamp(0, tw)
    tmax_y = tmax[..., 1].clamp(0, th)
    return tmin_x, tmin_y, tmax_x, tmax_y


def depth_bits(depths: Tensor) -> Tensor:
    return depths.detach().to(torch.float32).contiguous().view(torch.int32).to(torch.int64)


def isect_tiles(means2d: Tensor, radii: Tensor, depths: Tensor, ts: int, tw: int, th: int, sort: bool = True):
    """gsplat ``_isect_tiles``.  -> tiles_per_gauss int32 [C,N], isect_ids int64 [I], flatten_ids int32 [I]."""
    C, N = radii.shape
    tmin_x, tmin_y, tmax_x, tmax_y = tile_rects(means2d, radii, ts, tw, th)
    tpg = ((tmax_x - tmin_x) * (tmax_y - tmin_y) * (radii > 0)).to(torch.int64)
    counts = tpg.flatten()
    total = int(counts.sum())
    tb = (tw * th).bit_length()
    flat = torch.arange(C * N, dtype=torch.int64)
    idx = torch.repeat_interleave(flat, counts)
    starts = torch.cumsum(counts, 0) - counts
    local = torch.arange(total, dtype=torch.int64) - torch.repeat_interleave(starts, counts)
    w = (tmax_x - tmin_x).flatten()[idx]
    ty = tmin_y.flatten()[idx] + torch.div(local, w.clamp(min=1), rounding_mode="floor")
    tx = tmin_x.flatten()[idx] + local % w.clamp(min=1)
    tile_id = ty * tw + tx
    cam = torch.div(idx, N, rounding_mode="floor")
    dbits = depth_bits(depths).flatten()[idx]
    isect_ids = (cam << (32 + tb)) | (tile_id << 32) | (dbits & 0xFFFFFFFF)
    flatten_ids = idx.to(torch.int32)
    if sort:
        isect_ids, order = torch.sort(isect_ids, stable=True)
        flatten_ids = flatten_ids[order]
    return tpg.to(torch.int32), isect_ids, flatten_ids


def isect_offset_encode(isect_ids: Tensor, C: int, tw: int, th: int) -> Tensor:
    """gsplat ``_isect_offset_encode``: exclusive prefix of per-(camera,tile) counts, int32 [C,th,tw]."""
    tb = (tw * th).bit_length()
    counts = torch.zeros(C * th * tw, dtype=torch.int64)
    if isect_ids.numel() > 0:
        hi = isect_ids >> 32
        cam = hi >> tb
        tile = hi & ((1 << tb) - 1)
        counts.scatter_add_(0, cam * (tw * th) + tile, torch.ones_like(hi))
    offsets = torch.cumsum(counts, 0) - counts
    return offsets.reshape(C, th, tw).to(torch.int32)


# --------------------------------------------------------------------------
# A.6 compositing (definition of results = the CUDA kernel's per-pixel loop)
# --------------------------------------------------------------------------
def _composite(px: Tensor, py: Tensor, xy: Tensor, con: Tensor, op: Tensor, col: Tensor,
               extra_valid: Optional[Tensor] = None):
    """Front-to-back compositing of an ordered list of L Gaussians onto P pixels.

    Restates ``rasterize_to_pixels_fwd``: sigma = .5(a dx^2 + c dy^2) + b dx dy with
    dx = x_gauss - px; alpha = min(.999, o exp(-sigma)); skip if sigma < 0 or alpha < 1/255;
    stop (exclusive) at the first kept Gaussian with T(1-alpha) <= 1e-4.
    Returns (colors [P,D], T_final [P], weights [P,L], alive [P,L], ambiguous [P]) where
    ``ambiguous`` marks pixels that have, while still alive, a Gaussian within relative ``AMBIG_TOL`` of
    one of the three discontinuities (alpha = 1/255, T(1-alpha) = 1e-4, sigma = 0): a float32
    implementation may legitimately take the other branch there, changing the pixel by up to ~4e-3."""
    dx = xy[None, :, 0] - px[:, None]
    dy = xy[None, :, 1] - py[:, None]
    sigma = 0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) + con[None, :, 1] * dx * dy
    alpha = torch.clamp(op[None, :] * torch.exp(-sigma), max=ALPHA_MAX)
    valid = (sigma >= 0) & (alpha >= ALPHA_MIN)
    if extra_valid is not None:
        valid = valid & extra_valid
    a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_minus = 1.0 - a_eff
    T_incl = torch.cumprod(one_minus, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    stop = valid & (T_incl.detach() <= T_STOP)
    alive = torch.cumsum(stop.to(torch.int64), dim=1) == 0
    w = a_eff * T_excl * alive
    out = w @ col
    T_final = torch.prod(torch.where(alive, one_minus, torch.ones_like(one_minus)), dim=1)
    with torch.no_grad():
        raw = op[None, :] * torch.exp(-sigma)
        in_list = alive if extra_valid is None else (alive & extra_valid)
        near_alpha = (raw - ALPHA_MIN).abs() <= AMBIG_TOL * ALPHA_MIN
        near_sigma = (sigma.abs() <= AMBIG_TOL) & (raw >= ALPHA_MIN * (1 - AMBIG_TOL)) & ((dx != 0) | (dy != 0))
        near_stop = valid & ((T_incl - T_STOP).abs() <= AMBIG_TOL * T_STOP)
        # the Gaussian that stops a pixel is itself not "alive"; include it
        first_stop = stop & (torch.cumsum(stop.to(torch.int64), dim=1) == 1)
        ambiguous = (((near_alpha | near_sigma) & in_list) | (near_stop & (in_list | first_stop))).any(dim=1)
    return out, T_final, w, alive, ambiguous


def rasterize_to_pixels(
    means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor,
    width: int, height: int, ts: int, isect_offsets: Tensor, flatten_ids: Tensor,
    backgrounds: Optional[Tensor] = None, return_ambiguous: bool = False,
):
    """Tile-list compositor, semantics of gsplat's CUDA ``rasterize_to_pixels_fwd``.
    means2d [C,N,2], conics [C,N,3], colors [C,N,D], opacities [C,N]
    -> render_colors [C,H,W,D], render_alphas [C,H,W,1] (+ ambiguous [C,H,W] bool on request)."""
    C, N = means2d.shape[:2]
    D = colors.shape[-1]
    th, tw = isect_offsets.shape[1:]
    dt = means2d.dtype
    m2 = means2d.reshape(C * N, 2)
    cn = conics.reshape(C * N, 3)
    cl = colors.reshape(C * N, D)
    op = opacities.reshape(C * N)
    offs = isect_offsets.flatten().tolist() + [int(flatten_ids.numel())]
    rows = []
    arows = []
    amb = torch.zeros(C, height * width, dtype=torch.bool)
    for c in range(C):
        img = torch.zeros(height, width, D, dtype=dt)
        aimg = torch.zeros(height, width, 1, dtype=dt)
        pieces = []
        for tyi in range(th):
            for txi in range(tw):
                t = (c * th + tyi) * tw + txi
                s, e = offs[t], offs[t + 1]
                if e <= s:
                    continue
                y0, y1 = tyi * ts, min(tyi * ts + ts, height)
                x0, x1 = txi * ts, min(txi * ts + ts, width)
                yy, xx = torch.meshgrid(torch.arange(y0, y1), torch.arange(x0, x1), indexing="ij")
                px = xx.flatten().to(dt) + 0.5
                py = yy.flatten().to(dt) + 0.5
                ids = flatten_ids[s:e].to(torch.int64)
                out, T_final, _, _, ambig = _composite(px, py, m2[ids], cn[ids], op[ids], cl[ids])
                amb[c, (yy.flatten() * width + xx.flatten())] = ambig
                pieces.append((y0, y1, x0, x1, out, T_final))
        # assemble without in-place writes on tensors that need grad
        if pieces:
            idx_y = torch.cat([torch.arange(y0, y1).repeat_interleave(x1 - x0) for (y0, y1, x0, x1, _, _) in pieces])
            idx_x = torch.cat([torch.arange(x0, x1).repeat(y1 - y0) for (y0, y1, x0, x1, _, _) in pieces])
            vals = torch.cat([p[4] for p in pieces], dim=0)
            tf = torch.cat([p[5] for p in pieces], dim=0)
            lin = idx_y * width + idx_x
            img = img.reshape(height * width, D).index_put((lin,), vals).reshape(height, width, D)
            aimg = aimg.reshape(height * width, 1).index_put((lin,), (1.0 - tf)[:, None]).reshape(height, width, 1)
        if backgrounds is not None:
            img = img + (1.0 - aimg) * backgrounds[c][None, None, :]
        rows.append(img)
        arows.append(aimg)
    if return_ambiguous:
        return torch.stack(rows, 0), torch.stack(arows, 0), amb.reshape(C, height, width)
    return torch.stack(rows, 0), torch.stack(arows, 0)


def rasterize_bruteforce(
    means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor, radii: Tensor, depths: Tensor,
    width: int, height: int, ts: int, backgrounds: Optional[Tensor] = None,
):
    """Tile-free cross-check: every pixel walks ALL visible Gaussians of its camera in
    (depth bits, index) order; a Gaussian counts for a pixel only when the pixel's tile lies in the
    Gaussian's tile rectangle (that is what the tile lists encode).  Independent of isect_tiles'
    key construction, sort and offsets."""
    C, N = radii.shape
    D = colors.shape[-1]
    dt = means2d.dtype
    tw = (width + ts - 1) // ts
    th = (height + ts - 1) // ts
    tmin_x, tmin_y, tmax_x, tmax_y = tile_rects(means2d, radii, ts, tw, th)
    yy, xx = torch.meshgrid(torch.arange(height), torch.arange(width), indexing="ij")
    pxi = xx.flatten()
    pyi = yy.flatten()
    px = pxi.to(dt) + 0.5
    py = pyi.to(dt) + 0.5
    ptx = torch.div(pxi, ts, rounding_mode="floor")
    pty = torch.div(pyi, ts, rounding_mode="floor")
    outs, alphas = [], []
    for c in range(C):
        vis = torch.nonzero(radii[c] > 0).flatten()
        key = depth_bits(depths[c])[vis]
        order = torch.sort(key, stable=True).indices
        g = vis[order]
        in_rect = (
            (ptx[:, None] >= tmin_x[c, g][None]) & (ptx[:, None] < tmax_x[c, g][None])
            & (pty[:, None] >= tmin_y[c, g][None]) & (pty[:, None] < tmax_y[c, g][None])
        )
        out, T_final, _, _, _ = _composite(px, py, means2d[c, g], conics[c, g], opacities[c, g], colors[c, g], in_rect)
        a = (1.0 - T_final)[:, None]
        if backgrounds is not None:
            out = out + (1.0 - a) * backgrounds[c][None, :]
        outs.append(out.reshape(height, width, D))
        alphas.append(a.reshape(height, width, 1))
    return torch.stack(outs, 0), torch.stack(alphas, 0)


# --------------------------------------------------------------------------
# A.1 orchestration
# --------------------------------------------------------------------------
def rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor,
    viewmats: Tensor, Ks: Tensor, width: int, height: int,
    near_plane: float = 0.01, far_plane: float = 1e10, radius_clip: float = 0.0, eps2d: float = 0.3,
    sh_degree: Optional[int] = None, packed: bool = False, tile_size: int = 16,
    backgrounds: Optional[Tensor] = None, render_mode: str = "RGB", sparse_grad: bool = False,
    absgrad: bool = False, rasterize_mode: str = "classic", channel_chunk: int = 32,
    features: Optional[Tensor] = None, bruteforce: bool = False, radii_override: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    """gsplat ``rendering.rasterization`` (non-packed, non-distributed, pinhole).

    ``features`` is this build's extension (not in gsplat): [N,F] per-Gaussian feature channels
    composited in the same pass right after the RGB channels, i.e. output channels are
    [rgb(3) | features(F) | depth(1 if +D/+ED)].  ``bruteforce`` swaps the tile-list compositor
    for the tile-free cross-check."""
    N = means.shape[0]
    C = viewmats.shape[0]
    assert means.shape == (N, 3) and quats.shape == (N, 4) and scales.shape == (N, 3)
    assert opacities.shape == (N,)
    assert viewmats.shape == (C, 4, 4) and Ks.shape == (C, 3, 3)
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED")
    assert not packed and not sparse_grad
    radii, means2d, depths, conics, comps = fully_fused_projection(
        means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
        calc_compensations=(rasterize_mode == "antialiased"), radii_override=radii_override,
    )
    opac = opacities[None, :].expand(C, N)
    if comps is not None:
        opac = opac * comps
    clamp_ambiguous = None
    if sh_degree is None:
        cols = colors[None].expand(C, -1, -1) if colors.dim() == 2 else colors
    else:
        camtoworlds = torch.inverse(viewmats)
        dirs = means[None, :, :] - camtoworlds[:, None, :3, 3]
        shs = colors[None].expand(C, -1, -1, -1) if colors.dim() == 3 else colors
        cols = spherical_harmonics(sh_degree, dirs, shs, masks=radii > 0)
        # clamp_min(. + 0.5, 0) is a step for the gradient: a colour within float32 resolution of the clamp may fall
        # either side in a float32 implementation; flagged per (Gaussian, colour channel) for the gradient tests
        clamp_ambiguous = ((cols.detach() + 0.5).abs() < 1e-6).any(dim=0)
        cols = torch.clamp_min(cols + 0.5, 0.0)
    if features is not None:
        f = features[None].expand(C, -1, -1) if features.dim() == 2 else features
        cols = torch.cat([cols, f], dim=-1)
    bgs = backgrounds
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[..., None]], dim=-1)
        if bgs is not None:
            bgs = torch.cat([bgs, torch.zeros(C, 1, dtype=bgs.dtype)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if bgs is not None:
            bgs = torch.zeros(C, 1, dtype=bgs.dtype)
    tw = math.ceil(width / tile_size)
    th = math.ceil(height / tile_size)
    tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, tile_size, tw, th)
    isect_offsets = isect_offset_encode(isect_ids, C, tw, th)
    if bruteforce:
        render_colors, render_alphas = rasterize_bruteforce(
            means2d, conics, cols, opac, radii, depths, width, height, tile_size, bgs)
        ambiguous = None
    else:
        render_colors, render_alphas, ambiguous = rasterize_to_pixels(
            means2d, conics, cols, opac, width, height, tile_size, isect_offsets, flatten_ids, bgs,
            return_ambiguous=True)
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat(
            [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1)
    meta = {
        "camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths,
        "conics": conics, "opacities": opac, "tile_width": tw, "tile_height": th,
        "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": isect_offsets, "width": width, "height": height, "tile_size": tile_size,
        "n_cameras": C, "ambiguous": ambiguous, "colour_clamp_ambiguous": clamp_ambiguous,
    }
    return render_colors, render_alphas, meta
