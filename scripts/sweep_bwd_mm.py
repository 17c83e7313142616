"""Backward compositor: the planner's vector / bf16 kernels against raster_bwd_mm_kernel (POGS_RAST_BWD_MFMA=5) over the
shapes POGS renders.  Prints HIP-event times of pr_raster_bwd per call."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering, _lib
from pogs_amd.scenes import make_scene
dev = torch.device("cuda:0")

def run(name, N, W, H, F, ts, sh, mode, geom_only, steps=10, colours_only=False):
    sc = make_scene(N, W, H, n_features=max(F, 1), seed=0)
    p = {k: v.to(dev) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
    for k, v in p.items():
        v.requires_grad_(k == "means" if geom_only else k in ("f", "sh") if colours_only else True)
    vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
    kw = dict(packed=False, tile_size=ts, absgrad=not geom_only)
    if sh:
        kw.update(sh_degree=3, render_mode=mode)
        D = 3 + F + (1 if mode == "RGB+ED" else 0)
    else:
        kw.update(sh_degree=None, render_mode="RGB")
        D = F
    vc, va = torch.randn(1, H, W, D, device=dev), torch.randn(1, H, W, 1, device=dev)
    def step():
        for t in p.values(): t.grad = None
        if sh:
            r, a, _ = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, W, H,
                                              features=p["f"] if F else None, **kw)
        else:
            r, a, _ = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["f"], vm, K, W, H, **kw)
        torch.autograd.backward([r, a], [vc, va])
    out = {}
    for variant, env in (("old", dict(POGS_RAST_BWD_MFMA=1)), ("planner", dict(POGS_RAST_BWD_MFMA=None)), ("mm", dict(POGS_RAST_BWD_MFMA=5)), ("mm split", dict(POGS_RAST_BWD_MFMA=6))):
        _lib.set_tuning(**env)
        for _ in range(3): step()
        timer = rendering.StageTimer(only=("pr_raster_bwd", "pr_raster_fwd"))
        rendering.set_stage_timer(timer)
        for _ in range(steps): step()
        torch.cuda.synchronize()
        rendering.set_stage_timer(None)
        s = timer.summary()
        out[variant] = round(s["pr_raster_bwd"][1] / steps, 4)
        out["fwd"] = round(s["pr_raster_fwd"][1] / steps, 4)
    _lib.set_tuning(POGS_RAST_BWD_MFMA=None)
    print(f"{name:46s} D={D:3d} ts={ts:2d} {json.dumps(out)}", flush=True)
    del p, vc, va
    torch.cuda.empty_cache()

only = sys.argv[1:] or None
cases = [
    ("cfg2 1M 1080p rgb+ed+32", 1000000, 1920, 1080, 32, 16, True, "RGB+ED", False),
    ("step call 1: 1M 1080p rgb+ed", 1000000, 1920, 1080, 0, 16, True, "RGB+ED", False),
    ("step call 2: 1M 960x540 96 features", 1000000, 960, 540, 96, 16, False, "RGB", False),
    ("step call 3: 1M 450x250 64 features", 1000000, 450, 250, 64, 16, False, "RGB", False),
    ("step call 3 as trained: colours only", 1000000, 450, 250, 64, 16, False, "RGB", False, 10, True),
    ("cfg4 3M 1080p rgb+ed+32", 3000000, 1920, 1080, 32, 16, True, "RGB+ED", False),
    ("cfg5 6M 4K rgb+ed+64", 6000000, 3840, 2160, 64, 16, True, "RGB+ED", False),
    ("tracker 200k 640x360 rgb+ed ts8, means only", 200000, 640, 360, 0, 8, True, "RGB+ED", True),
    ("tracker 200k 640x360 64 feat ts16, means only", 200000, 640, 360, 64, 16, False, "RGB", True),
    ("cfg1 10k 256x256 rgb+16", 10000, 256, 256, 16, 16, True, "RGB", False),
]
for c in cases:
    if only is None or any(o in c[0] for o in only):
        run(*c)
