"""A/B of the forward compositor: matrix-core variant against the vector-ALU kernel on one scene.
Usage: python scripts/ab_fwd_mfma.py N W H F   (prints max abs differences and the forward entry point's time)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pogs_amd import _lib, rendering  # noqa: E402
from pogs_amd.scenes import make_scene  # noqa: E402

dev = torch.device("cuda:0")
n, w, h, f = (int(v) for v in sys.argv[1:5])
sc = make_scene(n, w, h, n_features=max(f, 1), seed=0)
p = {k: v.to(dev) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh,
                                  f=sc.features).items()}
vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)


def render():
    with torch.no_grad():
        return rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, w, h,
                                       sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=16, features=p["f"])


res = {}
for mode in ("0", "1"):
    _lib.set_tuning(POGS_RAST_FWD_MFMA=mode)
    for _ in range(3):
        r, a, meta = render()
    t = rendering.StageTimer()
    rendering.set_stage_timer(t)
    for _ in range(10):
        r, a, meta = render()
    torch.cuda.synchronize()
    rendering.set_stage_timer(None)
    res[mode] = (r.clone(), a.clone(), t.summary()["pr_raster_fwd"][1] / 10)
r0, a0, t0 = res["0"]
r1, a1, t1 = res["1"]
d = (r0 - r1).abs()
print(json.dumps({"cfg": sys.argv[1:5], "fwd_ms_valu": round(t0, 4), "fwd_ms_mfma": round(t1, 4),
                  "max_abs_rgb": d[..., :3].max().item(), "max_abs_feat": d[..., 3:-1].max().item(),
                  "max_abs_depth": d[..., -1].max().item(), "alpha_equal": bool(torch.equal(a0, a1)),
                  "feat_scale": r0[..., 3:-1].abs().max().item()}))
