"""Backward compositor with frozen appearance (geometry gradients only) at the benchmark shape: vector kernels at
4 and 1 pixels per lane against the matrix-core variant.  Prints the entry point's device time per variant."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pogs_amd import _lib, rendering  # noqa: E402
from pogs_amd.scenes import make_scene  # noqa: E402

dev = torch.device("cuda:0")
n, w, h, f = 1_000_000, 1920, 1080, 32
sc = make_scene(n, w, h, n_features=f, seed=0)
p = {k: v.to(dev) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh,
                                  f=sc.features).items()}
vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
feats36 = torch.cat([p["f"], torch.rand(n, 4, device=dev)], -1).contiguous()   # 36 frozen channels


def step():
    m = p["means"].detach().requires_grad_(True)
    r, a, _ = rendering.rasterization(m, p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), feats36, vm, K, w, h,
                                      packed=False, tile_size=16)
    (r.square().mean() + a.mean()).backward()


out = {}
for name, env in (("vector 4 px/lane", {"POGS_RAST_BWD_PPL": "4", "POGS_RAST_BWD_MFMA": "0"}),
                  ("vector 2 px/lane", {"POGS_RAST_BWD_PPL": "2", "POGS_RAST_BWD_MFMA": "0"}),
                  ("vector 1 px/lane", {"POGS_RAST_BWD_PPL": "1", "POGS_RAST_BWD_MFMA": "0"}),
                  ("matrix 1 px/lane", {"POGS_RAST_BWD_PPL": "1", "POGS_RAST_BWD_MFMA": "1"})):
    _lib.set_tuning(**env)
    for _ in range(3):
        step()
    t = rendering.StageTimer()
    rendering.set_stage_timer(t)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    rendering.set_stage_timer(None)
    out[name] = round(t.summary()["pr_raster_bwd"][1] / 10, 4)
print(json.dumps(out))
