"""A/B of the geometry-only backward: matrix-core dot products (POGS_RAST_BWD_MFMA=1) against the vector kernel.
Usage: python scripts/ab_bwd_geom.py D   (prints NaN counts and max differences of the three geometry gradients)"""
import os, sys, math, torch
sys.path.insert(0, ".")
from pogs_amd import rendering as rd, _lib
from tests._util import small_scene, scene_args
gpu = torch.device("cuda:0")
sc = small_scene(n=600, w=70, h=45, views=2, seed=8, lo=0.02, hi=0.3)
a = scene_args(sc, device=gpu)
D = int(sys.argv[1])
radii, means2d, depths, conics, _ = rd.fully_fused_projection(a["means"], a["quats"], a["scales"], a["viewmats"], a["Ks"], sc.width, sc.height)
tw, th = math.ceil(sc.width / 16), math.ceil(sc.height / 16)
_, ids, flat = rd.isect_tiles(means2d, radii, depths, 16, tw, th)
offs = rd.isect_offset_encode(ids, 2, tw, th)
g = torch.Generator().manual_seed(0)
colors = torch.randn(2, sc.n, D, generator=g).to(gpu)
opac = a["opacities"][None].expand(2, -1).contiguous()
vc = torch.randn(2, sc.height, sc.width, D, generator=g).to(gpu)
res = {}
for mode in ("0", "1"):
    _lib.set_tuning(POGS_RAST_BWD_MFMA=mode)
    m = means2d.detach().clone().requires_grad_(True); c = conics.detach().clone().requires_grad_(True); o = opac.clone().requires_grad_(True)
    out, al = rd.rasterize_to_pixels(m, c, colors, o, sc.width, sc.height, 16, offs, flat, absgrad=True)
    ((out * vc).sum() + al.sum()).backward()
    res[mode] = (m.grad.clone(), c.grad.clone(), o.grad.clone())
for i, name in enumerate(("means2d", "conics", "opac")):
    x0, x1 = res["0"][i], res["1"][i]
    print(name, "nan:", torch.isnan(x1).sum().item(), "of", x1.numel(), "maxdiff(finite):", (x0 - x1)[~torch.isnan(x1)].abs().max().item(), "scale", x0.abs().max().item())
