"""Compositor times per channel count and pixels per lane at the benchmark geometry (1M Gaussians, 1080p):
one fused forward+backward per setting, HIP-event stage times.  usage: sweep_ppl.py [F ...]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering, _lib
from pogs_amd.scenes import make_scene
dev = torch.device("cuda:0")
N, W, H = [int(x) for x in os.environ.get('SWEEP_SHAPE', '1000000,1920,1080').split(',')]
Fs = [int(a) for a in sys.argv[1:]] or [0, 4, 12, 20, 28, 32]
for F in Fs:
    sc = make_scene(N, W, H, n_features=max(F, 1), seed=0)
    p = {k: v.to(dev).requires_grad_(True) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
    vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
    D = 3 + F + 1
    vc, va = torch.randn(1, H, W, D, device=dev), torch.randn(1, H, W, 1, device=dev)
    def step():
        for t in p.values(): t.grad = None
        r, a, _ = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, W, H, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=16, absgrad=True, features=p["f"] if F else None)
        torch.autograd.backward([r, a], [vc, va])
    row = {}
    for ppl in (4, 2, 1):
        _lib.set_tuning(POGS_RAST_FWD_PPL=ppl, POGS_RAST_BWD_PPL=ppl)
        for _ in range(5): step()
        timer = rendering.StageTimer(only=("pr_raster_bwd", "pr_raster_fwd"))
        rendering.set_stage_timer(timer)
        for _ in range(10): step()
        torch.cuda.synchronize()
        rendering.set_stage_timer(None)
        s = timer.summary()
        row[ppl] = {k: round(v[1] / v[0], 4) for k, v in s.items()}
    print("D", D, json.dumps(row))
_lib.set_tuning(POGS_RAST_FWD_PPL=None, POGS_RAST_BWD_PPL=None)
