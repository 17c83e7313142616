"""Secondary workloads of BASELINE.json (single GPU): cfg1 plumbing shape, cfg3 tracking loop, cfg5-like
stress shape.  Prints one JSON object per config.  The headline (cfg2) lives in bench.py."""
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import mlp, rendering, tracking  # noqa: E402
from pogs_amd.scenes import make_scene  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, iters, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def quat_mul(a, b):
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], -1)


def quat_rotate(q, v):
    qv = torch.cat([torch.zeros_like(v[..., :1]), v], -1)
    qc = q * torch.tensor([1.0, -1.0, -1.0, -1.0], device=q.device)
    return quat_mul(quat_mul(q, qv), qc)[..., 1:]


def training_shape(n, w, h, f, ts=16, iters=10, warmup=3):
    sc = make_scene(n, w, h, n_features=f, seed=0)
    p = {k: v.to(dev).requires_grad_(True) for k, v in dict(means=sc.means, quats=sc.quats, ls=sc.log_scales,
                                                            ol=sc.opacity_logits, sh=sc.sh, f=sc.features).items()}
    vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
    D = 3 + f + 1
    g = torch.Generator().manual_seed(1)
    vc, va = torch.randn(1, h, w, D, generator=g).to(dev), torch.randn(1, h, w, 1, generator=g).to(dev)
    info = {}

    def step():
        for t in p.values():
            t.grad = None
        r, a, meta = rendering.rasterization(p["means"], p["quats"], p["ls"].exp(), torch.sigmoid(p["ol"]), p["sh"], vm,
                                             K, w, h, sh_degree=3, render_mode="RGB+ED", packed=False, absgrad=True,
                                             tile_size=ts, features=p["f"])
        torch.autograd.backward([r, a], [vc, va])
        info["I"] = meta["n_composited_isects"]

    ms = timed(step, iters, warmup)
    return {"gaussians": n, "size": [w, h], "channels": D, "tile": ts, "n_isects": info["I"], "ms_fwd_bwd": ms,
            "Mpix_s": w * h / ms / 1e3}


def tracking_loop(n=200_000, w=640, h=360, groups=4, iters=50, frames=3):
    """cfg3: per frame `iters` pose-optimisation iterations; each renders RGB+ED (SH degree 3, tile 8) and a
    64-channel feature image (tile 16, geometry other than the means detached) of the object's Gaussians
    after a per-group rigid transform, pushes the feature image through the frozen 4-layer channel MLP
    (pogs/pogs.py:516-520; fused kernel, POGS_BENCH_TORCH_MLP=1 for the torch operators), back-propagates an L2
    loss to the [G,7] pose deltas, Adam step.
    The rigid transform (the reference's Warp kernel, pogs/tracking/utils.py:53-93) is pr_rigid_fwd/bwd; set
    POGS_BENCH_TORCH_RIGID=1 to time the plain-torch formulation (one-hot GEMM gather) instead."""
    sc = make_scene(n, w, h, n_features=64, seed=0)
    means, quats = sc.means.to(dev), torch.nn.functional.normalize(sc.quats, dim=-1).to(dev)
    scales, opac = sc.log_scales.exp().to(dev), torch.sigmoid(sc.opacity_logits).to(dev)
    sh, feats = sc.sh.to(dev), sc.features.to(dev)
    vm, K = sc.viewmats.to(dev), sc.Ks.to(dev)
    labels = (torch.arange(n, device=dev) % groups)
    onehot = torch.nn.functional.one_hot(labels, groups).float()  # gather as a GEMM: its backward is a GEMM too
    deltas = torch.zeros(groups, 7, device=dev)
    deltas[:, 3] = 1.0
    p2w = deltas.clone()
    p2w[:, :3] = torch.randn(groups, 3, device=dev) * 0.3
    labels32 = labels.int()
    torch_rigid = os.environ.get("POGS_BENCH_TORCH_RIGID") == "1"
    torch_mlp = os.environ.get("POGS_BENCH_TORCH_MLP") == "1"
    net = mlp.FeatureMLP(device=dev)
    for prm in net.parameters():
        prm.requires_grad_(False)

    def project(f, a):
        if not torch_mlp:
            return net(f, a)
        x = torch.where(a > 0, f / a.clamp(min=1e-6).detach(), torch.zeros(64, device=dev))
        return torch.nn.Sequential.forward(net, x.view(-1, 64)).view(f.shape)
    deltas.requires_grad_(True)
    opt = torch.optim.Adam([deltas], lr=1e-3)
    with torch.no_grad():
        tgt_rgb, _, _ = rendering.rasterization(means, quats, scales, opac, sh, vm, K, w, h, sh_degree=3,
                                                render_mode="RGB+ED", packed=False, tile_size=8)
        tgt_f, tgt_a, _ = rendering.rasterization(means, quats, scales, opac, feats, vm, K, w, h, packed=False)
        tgt_f = project(tgt_f, tgt_a)

    def iteration():
        opt.zero_grad(set_to_none=True)
        if torch_rigid:
            per = onehot @ torch.cat([deltas[:, :3], torch.nn.functional.normalize(deltas[:, 3:], dim=-1)], -1)
            dq = per[:, 3:]
            m = quat_rotate(dq, means) + per[:, :3]
            q = quat_mul(dq, quats)
        else:
            m, q = tracking.apply_to_model(p2w, deltas, labels32, means, quats)
        rgb, _, _ = rendering.rasterization(m, q, scales, opac, sh, vm, K, w, h, sh_degree=3, render_mode="RGB+ED",
                                            packed=False, tile_size=8)
        f, fa, _ = rendering.rasterization(m, q.detach(), scales, opac, feats, vm, K, w, h, packed=False)
        loss = (rgb - tgt_rgb).square().mean() + (project(f, fa) - tgt_f).square().mean()
        loss.backward()
        opt.step()

    def frame():
        for _ in range(iters):
            iteration()

    ms = timed(frame, frames, 1)
    return {"gaussians": n, "size": [w, h], "iters_per_frame": iters, "ms_per_frame": ms, "ms_per_iter": ms / iters,
            "rigid": "torch" if torch_rigid else "pr_rigid", "mlp": "torch" if torch_mlp else "pr_mlp"}


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg3", "cfg5"]
    if "cfg1" in which:
        print(json.dumps({"cfg1": training_shape(10_000, 256, 256, 16, iters=50, warmup=10)}), flush=True)
    if "cfg3" in which:
        print(json.dumps({"cfg3": tracking_loop()}), flush=True)
    if "cfg5" in which:
        print(json.dumps({"cfg5_one_view": training_shape(6_000_000, 3840, 2160, 64, iters=5, warmup=2)}), flush=True)
