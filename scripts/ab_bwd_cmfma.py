"""A/B of the backward compositor at the benchmark geometry (1M Gaussians, 1080p, 36 channels): the vector kernel
against the kernel with the deferred matrix-core colour gradient (POGS_RAST_BWD_MFMA=3); HIP-event stage times and
the largest difference between the two sets of gradients.  usage: ab_bwd_cmfma.py [N W H F]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering, _lib
from pogs_amd.scenes import make_scene
dev = torch.device("cuda:0")
N, W, H, F = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (1000000, 1920, 1080, 32)
sc = make_scene(N, W, H, n_features=F, seed=0)
p = {k: v.to(dev).requires_grad_(True) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
D = 3 + F + 1
vc, va = torch.randn(1, H, W, D, device=dev), torch.randn(1, H, W, 1, device=dev)
def step():
    for t in p.values(): t.grad = None
    r, a, _ = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, W, H, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=16, absgrad=True, features=p["f"])
    torch.autograd.backward([r, a], [vc, va])
out, grads = {}, {}
VARIANTS = os.environ.get("AB_VARIANTS", "").split(",") if os.environ.get("AB_VARIANTS") else None
for name, env in (("vector", dict(POGS_RAST_BWD_MFMA=1, POGS_RAST_BWD_PPL=None)), ("mm", dict(POGS_RAST_BWD_MFMA=5, POGS_RAST_BWD_PPL=None)), ("deferred", dict(POGS_RAST_BWD_MFMA=3, POGS_RAST_BWD_PPL=2)),
                  ("split", dict(POGS_RAST_BWD_MFMA=4, POGS_RAST_BWD_PPL=None)),
                  ("split, two streams", dict(POGS_RAST_BWD_MFMA=4, POGS_RAST_BWD_PPL=None, POGS_RAST_BWD_SPLIT=1)),
                  ("two streams, vector colour-only kernel", dict(POGS_RAST_BWD_MFMA=1, POGS_RAST_BWD_PPL=None, POGS_RAST_BWD_SPLIT=1)), ("vector again", dict(POGS_RAST_BWD_MFMA=1, POGS_RAST_BWD_PPL=None, POGS_RAST_BWD_SPLIT=None))):
    if VARIANTS and name not in VARIANTS: continue
    _lib.set_tuning(**env)
    for _ in range(5): step()
    timer = rendering.StageTimer(only=("pr_raster_bwd", "pr_raster_fwd"))
    rendering.set_stage_timer(timer)
    for _ in range(20): step()
    torch.cuda.synchronize()
    rendering.set_stage_timer(None)
    out[name] = {k: round(v[1] / v[0], 4) for k, v in timer.summary().items()}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record()
    torch.cuda.synchronize()
    out[name]["ms_per_step"] = round(e0.elapsed_time(e1) / 20, 4)
    grads[name] = {k: t.grad.clone() for k, t in p.items()}
    print(name, json.dumps(out[name]), flush=True)
for k in p:
    a = grads["vector"][k]
    print(k, "max |diff| / max |ref| =", {n: float((a - grads[n][k]).abs().max() / a.abs().max()) for n in grads if n != "vector"})
_lib.set_tuning(POGS_RAST_BWD_MFMA=None, POGS_RAST_BWD_PPL=None)
