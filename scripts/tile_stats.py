"""Diagnostic: per-tile work distribution of the compositing kernels at the benchmark shape."""
import sys, math, torch
sys.path.insert(0, '.')
from pogs_amd import rendering
from pogs_amd.scenes import make_scene

dev = torch.device('cuda:0')
N, W, H, F = 1_000_000, 1920, 1080, 32
sc = make_scene(N, W, H, n_features=F, seed=0)
a = dict(means=sc.means, quats=sc.quats, scales=sc.log_scales.exp(), opacities=torch.sigmoid(sc.opacity_logits),
         colors=sc.sh, viewmats=sc.viewmats, Ks=sc.Ks, features=sc.features)
a = {k: v.to(dev) for k, v in a.items()}

class Ctx:
    def save_for_backward(self, *t): self.saved = t
    needs_input_grad = [False] * 11

with torch.no_grad():
    radii, means2d, depths, conics, _ = rendering.fully_fused_projection(a['means'], a['quats'], a['scales'], a['viewmats'], a['Ks'], W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = rendering.isect_tiles(means2d, radii, depths, 16, tw, th)
    offs = rendering.isect_offset_encode(ids, 1, tw, th)
    cols = torch.randn(1, N, 36, device=dev)
    ctx = Ctx()
    out, alpha = rendering._Rasterize.forward(ctx, means2d, conics, cols, a['opacities'][None].contiguous(), None, W, H, 16, offs, flat, False)
    last = ctx.saved[-1]
o = offs.flatten().long()
ends = torch.cat([o[1:], torch.tensor([ids.numel()], device=dev)])
length = (ends - o).float()
lt = last[0].view(th if False else H // 1, W)  # [H,W]
# per tile max last
pad_h, pad_w = th * 16, tw * 16
lp = torch.zeros(pad_h, pad_w, dtype=torch.int64, device=dev)
lp[:H, :W] = last[0].long()
tmax = lp.view(th, 16, tw, 16).amax(dim=(1, 3)).flatten()
proc = (tmax - o).clamp(min=0).float()
print('tiles', o.numel(), 'I', ids.numel(), 'visible', int((radii > 0).sum()))
for name, v in (('list length', length), ('processed (bwd range)', proc)):
    q = torch.quantile(v, torch.tensor([0.5, 0.9, 0.99, 1.0], device=dev))
    print(name, 'sum', v.sum().item(), 'mean', v.mean().item(), 'p50/p90/p99/max', q.tolist(), 'empty frac', (v == 0).float().mean().item())
# longest-processing-time bound with 2048 concurrent waves
s = proc.sort(descending=True).values
print('max/mean ratio', (s[0] / proc.mean()).item(), ' ideal makespan share of max tile: ', (s[0] / (proc.sum() / 2048)).item())
