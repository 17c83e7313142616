"""Generates the golden vectors under tests/golden/ from the float64 CPU oracle.

    python -m tests.golden.make_golden

The reference holds no vectors for the rasterizer ("parity unpinned"); these files pin THIS build's
oracle so later rounds cannot drift silently.  Each .npz holds the inputs, the float64 outputs, the
binning integers, the ambiguity mask, the cotangents and the parameter gradients of one small case.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gsplat_ref as R  # noqa: E402
from pogs_amd.scenes import make_scene  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "rgb_ed_ts16": dict(n=256, w=64, h=64, f=0, views=1, seed=11, ts=16, mode="RGB+ED", sh_degree=3, antialiased=0),
    "rgb_ed_feat16_ts16": dict(n=256, w=64, h=48, f=16, views=1, seed=12, ts=16, mode="RGB+ED", sh_degree=3,
                               antialiased=0),
    "feat32_ts16_2views": dict(n=256, w=48, h=40, f=32, views=2, seed=13, ts=16, mode="RGB", sh_degree=-1,
                               antialiased=0),
    "rgb_ed_ts8_tracking": dict(n=200, w=56, h=42, f=0, views=1, seed=14, ts=8, mode="RGB+ED", sh_degree=3,
                                antialiased=0),
    "feat64_ts16": dict(n=200, w=40, h=30, f=64, views=1, seed=15, ts=16, mode="RGB", sh_degree=-1, antialiased=0),
    "rgb_antialiased": dict(n=256, w=64, h=64, f=0, views=1, seed=16, ts=16, mode="RGB", sh_degree=2, antialiased=1),
}


def inputs_for(spec):
    sc = make_scene(spec["n"], spec["w"], spec["h"], n_features=max(spec["f"], 1), n_views=spec["views"],
                    seed=spec["seed"], scale_lo=0.02, scale_hi=0.3)
    return dict(means=sc.means, quats=sc.quats, scales=sc.log_scales.exp(), opacities=torch.sigmoid(sc.opacity_logits),
                sh=sc.sh, features=sc.features, viewmats=sc.viewmats, Ks=sc.Ks)


def call_kwargs(spec, inp, dtype, device="cpu", requires_grad=True):
    """Keyword arguments of the rasterization() call a case stands for (oracle and product share them)."""
    t = {k: v.to(dtype).to(device).contiguous() for k, v in inp.items()}
    if requires_grad:
        for k in t:
            if k != "Ks":
                t[k].requires_grad_(True)
    kw = dict(means=t["means"], quats=t["quats"], scales=t["scales"], opacities=t["opacities"],
              viewmats=t["viewmats"], Ks=t["Ks"], width=spec["w"], height=spec["h"], tile_size=spec["ts"],
              render_mode=spec["mode"], packed=False, absgrad=True,
              rasterize_mode="antialiased" if spec["antialiased"] else "classic")
    if spec["sh_degree"] >= 0:
        kw.update(colors=t["sh"], sh_degree=spec["sh_degree"])
        if spec["f"] > 0:
            kw.update(features=t["features"])
    else:
        kw.update(colors=t["features"], sh_degree=None)
    return t, kw


def cotangents(spec, shape_c, shape_a):
    g = torch.Generator().manual_seed(1000 + spec["seed"])
    return (torch.randn(shape_c, generator=g, dtype=torch.float32).double(),
            torch.randn(shape_a, generator=g, dtype=torch.float32).double())


def run_case(spec, dtype, inp=None, ambiguous=None):
    inp = inputs_for(spec) if inp is None else inp
    t, kw = call_kwargs(spec, inp, dtype)
    out, alpha, meta = R.rasterization(**kw)
    amb = meta["ambiguous"] if ambiguous is None else ambiguous
    vc, va = cotangents(spec, out.shape, alpha.shape)
    keep = (~amb)[..., None].to(torch.float64)
    vc, va = vc * keep, va * keep
    ((out * vc.to(dtype)).sum() + (alpha * va.to(dtype)).sum()).backward()
    colors_key = "sh" if spec["sh_degree"] >= 0 else "features"
    res = dict(render_colors=out.detach(), render_alphas=alpha.detach(), ambiguous=amb, v_render_colors=vc,
               v_render_alphas=va, isect_ids=meta["isect_ids"], flatten_ids=meta["flatten_ids"],
               isect_offsets=meta["isect_offsets"], radii=meta["radii"], tiles_per_gauss=meta["tiles_per_gauss"],
               v_means=t["means"].grad, v_quats=t["quats"].grad, v_scales=t["scales"].grad,
               v_opacities=t["opacities"].grad, v_colors=t[colors_key].grad, v_viewmats=t["viewmats"].grad)
    if spec["sh_degree"] >= 0 and spec["f"] > 0:
        res["v_features"] = t["features"].grad
    return res


def main():
    for name, spec in CASES.items():
        inp = inputs_for(spec)
        res = run_case(spec, torch.float64, inp)
        arrays = {f"spec_{k}": np.array(v) for k, v in spec.items()}
        arrays.update({f"in_{k}": v.numpy() for k, v in inp.items()})
        arrays.update({k: (v.float() if k.startswith("v_render") else v).numpy() for k, v in res.items()})
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(name, os.path.getsize(path) // 1024, "KiB", "isects", res["isect_ids"].numel(), "ambiguous",
              int(res["ambiguous"].sum()))


if __name__ == "__main__":
    main()
