"""The three backward variants at the benchmark geometry (1M Gaussians, 1080p, 36 channels), a few steps each, for
counter collection: run under `rocprofv3 --pmc ... -- python3 scripts/profile_bwd_variants.py`; the kernels of interest
are raster_bwd_kernel<16,2,36,true,true>, raster_bwd_cmfma_kernel<36>, raster_bwd_kernel<16,2,36,true,false> and
raster_bwd_colour_mfma_kernel<3>."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering, _lib
from pogs_amd.scenes import make_scene
dev = torch.device("cuda:0")
N, W, H, F = 1000000, 1920, 1080, 32
sc = make_scene(N, W, H, n_features=F, seed=0)
p = {k: v.to(dev).requires_grad_(True) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
D = 3 + F + 1
vc, va = torch.randn(1, H, W, D, device=dev), torch.randn(1, H, W, 1, device=dev)
for env in (dict(POGS_RAST_BWD_MFMA=1), dict(POGS_RAST_BWD_MFMA=3), dict(POGS_RAST_BWD_MFMA=4)):
    _lib.set_tuning(**env)
    for _ in range(3):
        for t in p.values(): t.grad = None
        r, a, _ = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, W, H, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=16, absgrad=True, features=p["f"])
        torch.autograd.backward([r, a], [vc, va])
    torch.cuda.synchronize()
_lib.set_tuning(POGS_RAST_BWD_MFMA=None)
