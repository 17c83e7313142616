"""Per-entry-point device times for one render configuration (diagnostic)."""
import sys, json, torch
sys.path.insert(0, ".")
from pogs_amd import rendering
from pogs_amd.scenes import make_scene
dev = torch.device("cuda:0")
n, w, h, f, ts, mode = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
sc = make_scene(n, w, h, n_features=max(f, 1), seed=0)
p = {k: v.to(dev).requires_grad_(True) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
def step():
    for t in p.values(): t.grad = None
    if mode == "rgb":
        r, a, meta = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, w, h, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=ts)
    elif mode == "feat":
        r, a, meta = rendering.rasterization(p["means"], p["quats"].detach(), p["ls"].exp().detach(), torch.sigmoid(p["ol"]).detach(), p["f"], vm, K, w, h, packed=False, tile_size=ts)
    else:
        r, a, meta = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, w, h, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=ts, features=p["f"])
    (r.square().mean() + a.mean()).backward()
    return meta
for _ in range(3): meta = step()
t = rendering.StageTimer(); rendering.set_stage_timer(t)
torch.cuda.synchronize()
import time; t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps({"cfg": sys.argv[1:], "I": meta["n_composited_isects"], "ms": round(el, 3), "stages": {k: round(v[1] / 10, 4) for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1][1])}}))
