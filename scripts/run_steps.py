"""A few fwd+bwd steps at the benchmark geometry (1M Gaussians, 1080p, RGB+ED + 32 features) with whatever kernel
variants the POGS_RAST_* environment selects; a target for `rocprofv3 --pmc ... -- python3 scripts/run_steps.py [steps]`."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering
from pogs_amd.scenes import make_scene
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, W, H, F = [int(a) for a in sys.argv[2:6]] if len(sys.argv) >= 6 else (1000000, 1920, 1080, 32)
sc = make_scene(N, W, H, n_features=F, seed=0)
p = {k: v.to(dev).requires_grad_(True) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales, ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
D = 3 + F + 1
vc, va = torch.randn(1, H, W, D, device=dev), torch.randn(1, H, W, 1, device=dev)
for _ in range(steps):
    for t in p.values(): t.grad = None
    r, a, _ = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm, K, W, H, sh_degree=3, render_mode="RGB+ED", packed=False, tile_size=16, absgrad=True, features=p["f"] if F else None)
    torch.autograd.backward([r, a], [vc, va])
torch.cuda.synchronize()
