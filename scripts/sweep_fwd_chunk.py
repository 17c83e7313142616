"""Forward compositor time against the width of its channel passes (POGS_RAST_FWD_CHUNK) and the pixels per lane
(POGS_RAST_FWD_PPL) on the small-image / wide-channel renders of POGS.  usage: sweep_fwd_chunk.py [steps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import _lib, rendering
from pogs_amd.scenes import make_scene

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
shapes = [("tracker 200k 640x360 64ch", 200_000, 640, 360, 1.0, 64), ("clip 1M 960x540 96ch", 1_000_000, 960, 540, 0.5, 96),
          ("dino 1M 450x250 64ch", 1_000_000, 450, 250, 450 / 1920, 64), ("cfg2 1M 1080p 36ch", 1_000_000, 1920, 1080, 1.0, 36)]
for name, n, W, H, ks, D in shapes:
    sc = make_scene(n, int(round(W / ks)), int(round(H / ks)), n_features=4, n_views=1, seed=0)
    K = sc.Ks[:1].clone()
    K[:, :2, :] *= ks
    g = torch.Generator().manual_seed(1)
    cols = torch.randn(n, D, generator=g).to(dev)
    args = dict(means=sc.means.to(dev), quats=sc.quats.to(dev), scales=sc.log_scales.exp().to(dev),
                opacities=torch.sigmoid(sc.opacity_logits).to(dev), colors=cols, viewmats=sc.viewmats[:1].to(dev), Ks=K.to(dev),
                width=W, height=H, tile_size=16, packed=False)
    res = []
    for ppl in ((None,) if os.environ.get("POGS_RAST_FWD_BLOCKS") else (None, 1, 2)):
        for chunk in (None, 8, 16, 24, 32, 48):
            if chunk is not None and chunk >= D and chunk != 48:
                continue
            _lib.set_tuning(POGS_RAST_FWD_CHUNK=chunk, POGS_RAST_FWD_PPL=ppl)
            try:
                with torch.no_grad():
                    for _ in range(3):
                        rendering.rasterization(**args)
                    timer = rendering.StageTimer(only=("pr_raster_fwd",))
                    rendering.set_stage_timer(timer)
                    for _ in range(steps):
                        rendering.rasterization(**args)
                    torch.cuda.synchronize()
                    rendering.set_stage_timer(None)
                c, ms = timer.summary()["pr_raster_fwd"]
                res.append((ppl, chunk, round(ms / steps, 4)))
            except Exception as exc:
                res.append((ppl, chunk, type(exc).__name__))
    print(name, res, flush=True)
_lib.set_tuning(POGS_RAST_FWD_CHUNK=None, POGS_RAST_FWD_PPL=None)
