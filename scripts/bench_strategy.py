"""Time the densification step (per-step statistics and one refinement) against the torch operator sequence the
published DefaultStrategy issues, at the benchmark's Gaussian count.

Usage: python scripts/bench_strategy.py [--n 1000000] [--reps 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pogs_amd.strategy import DefaultStrategy  # noqa: E402


def make(n, dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g, device=dev)          # noqa: E731
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)        # noqa: E731
    params = {"means": rn(n, 3), "scales": torch.log(r(n, 3) * 0.03 + 1e-3), "quats": rn(n, 4),
              "opacities": rn(n, 1) * 3, "features_dc": r(n, 3), "features_rest": r(n, 15, 3), "dino_feats": rn(n, 32)}
    params = {k: torch.nn.Parameter(v) for k, v in params.items()}
    opts = {k: torch.optim.Adam([v], lr=1e-3, eps=1e-15) for k, v in params.items()}
    for k, v in params.items():
        opts[k].state[v] = {"step": torch.tensor(3.0), "exp_avg": torch.zeros_like(v) + 0.1,
                            "exp_avg_sq": torch.zeros_like(v) + 0.1}
    stats = {"grad2d": r(n) * 4e-3, "count": torch.randint(0, 5, (n,), generator=g, device=dev).float()}
    return params, opts, stats


def torch_update_state(state, grads, radii, w, h, ncam):
    g = grads.clone()
    g[..., 0] *= w / 2.0 * ncam
    g[..., 1] *= h / 2.0 * ncam
    sel = radii > 0.0
    ids = torch.where(sel)[1]
    state["grad2d"].index_add_(0, ids, g[sel].norm(dim=-1))
    state["count"].index_add_(0, ids, torch.ones_like(ids, dtype=torch.float32))


def torch_refine(params, opts, state, strat, step):
    """duplicate -> split -> remove with device index operators, each materialising every tensor."""
    def rebuild(fp, fs):
        for name in list(params):
            old = params[name]
            new = torch.nn.Parameter(fp(name, old.detach()))
            st = opts[name].state.pop(old)
            for key in ("exp_avg", "exp_avg_sq"):
                st[key] = fs(st[key])
            opts[name].param_groups[0]["params"] = [new]
            opts[name].state[new] = st
            params[name] = new
        for k, v in state.items():
            if isinstance(v, torch.Tensor):
                state[k] = fp("state", v)
    dev = params["means"].device
    grads = state["grad2d"] / state["count"].clamp_min(1)
    high = grads > strat.grow_grad2d
    small = torch.exp(params["scales"]).max(dim=-1).values <= strat.grow_scale3d
    is_dup, is_split = high & small, high & ~small
    n_dup, n_split = is_dup.sum().item(), is_split.sum().item()
    sel = torch.where(is_dup)[0]
    rebuild(lambda n, p: torch.cat([p, p[sel]]), lambda v: torch.cat([v, torch.zeros((len(sel), *v.shape[1:]), device=dev)]))
    is_split = torch.cat([is_split, torch.zeros(n_dup, dtype=torch.bool, device=dev)])
    sel, rest = torch.where(is_split)[0], torch.where(~is_split)[0]
    scales = torch.exp(params["scales"][sel])
    q = torch.nn.functional.normalize(params["quats"][sel], dim=-1)
    w, x, y, z = q.unbind(-1)
    rot = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                       1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                       1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    samples = torch.einsum("nij,nj,bnj->bni", rot, scales, torch.randn(2, len(scales), 3, device=dev))

    def fp(name, p):
        twice = [2] + [1] * (p.dim() - 1)
        if name == "means":
            kids = (p[sel] + samples).reshape(-1, 3)
        elif name == "scales":
            kids = torch.log(scales / 1.6).repeat(2, 1)
        else:
            kids = p[sel].repeat(twice)
        return torch.cat([p[rest], kids])
    rebuild(fp, lambda v: torch.cat([v[rest], torch.zeros((2 * len(sel), *v.shape[1:]), device=dev)]))
    gone = torch.sigmoid(params["opacities"].flatten()) < strat.prune_opa
    if step > strat.reset_every:
        gone |= torch.exp(params["scales"]).max(dim=-1).values > strat.prune_scale3d
    n_prune = gone.sum().item()
    keep = torch.where(~gone)[0]
    rebuild(lambda n, p: p[keep], lambda v: v[keep])
    return n_dup, n_split, n_prune


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n, w, h = a.n, 1920, 1080
    strat = DefaultStrategy(grow_grad2d=0.0008, grow_scale3d=0.015, prune_opa=0.05, prune_scale3d=0.028, absgrad=True)
    grads = torch.randn(1, n, 2, device=dev) * 1e-4
    radii = torch.randint(0, 30, (1, n), device=dev, dtype=torch.int32)
    m2d = torch.zeros(1, n, 2, device=dev)
    m2d.absgrad = grads
    info = {"width": w, "height": h, "n_cameras": 1, "radii": radii, "means2d": m2d}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    params, opts, stats = make(n, dev)
    state = strat.initialize_state()
    state.update(grad2d=stats["grad2d"].clone(), count=stats["count"].clone())
    t_acc = timed(lambda: strat._update_state(params, state, info), 50)
    tstate = {"grad2d": stats["grad2d"].clone(), "count": stats["count"].clone()}
    t_acc_t = timed(lambda: torch_update_state(tstate, grads, radii, w, h, 1), 50)
    print(f"N = {n}")
    print(f"per-step statistics   fused {t_acc:.3f} ms ({n * 28 / t_acc / 1e6:.0f} GB/s)   torch {t_acc_t:.3f} ms   x{t_acc_t / t_acc:.1f}")

    def one(fn):
        ts = []
        for rep in range(a.reps):
            params, opts, stats = make(n, dev, seed=rep)
            state = {"grad2d": stats["grad2d"], "count": stats["count"], "scene_scale": 1.0}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = fn(params, opts, state)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return min(ts[1:]) if len(ts) > 1 else ts[0], res, len(params["means"])

    t_f, res_f, n_f = one(lambda p, o, s: strat._refine(p, o, s, 3600))
    t_t, res_t, n_t = one(lambda p, o, s: torch_refine(p, o, s, strat, 3600))
    floats = sum(v[0].numel() for v in params.values())
    print(f"refinement {res_f} -> {n_f} Gaussians   fused {t_f:.2f} ms "
          f"({n_f * floats * 3 * 8 / t_f / 1e6:.0f} GB/s of gathered rows)   torch {t_t:.2f} ms {res_t} -> {n_t}   x{t_t / t_f:.1f}")


if __name__ == "__main__":
    main()
