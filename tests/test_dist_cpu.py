"""world_size-2 gloo test of the camera-sharded step: flat gradient buffer + mean all-reduce must equal the
single-process mean over views.  The per-view "render" is the CPU oracle on a tiny scene (checker only)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pogs_amd.dist import FlatGrads, GradExchange, sharded_step, view_for_rank
from tests._util import small_scene


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(sc):
    p = dict(means=sc.means.double(), quats=sc.quats.double(), log_scales=sc.log_scales.double(),
             opacity_logits=sc.opacity_logits.double(), features=sc.features.double())
    return {k: v.clone().requires_grad_(True) for k, v in p.items()}


def _render_view(sc, p, v):
    from oracle import gsplat_ref as R

    out, alpha, _ = R.rasterization(p["means"], p["quats"], p["log_scales"].exp(), torch.sigmoid(p["opacity_logits"]),
                                    p["features"], sc.viewmats[v:v + 1].double(), sc.Ks[v:v + 1].double(), sc.width,
                                    sc.height, sh_degree=None, tile_size=16)
    w = torch.linspace(0.5, 1.5, out.numel(), dtype=torch.float64).view_as(out) * (1 + v)
    ((out * w).sum() + alpha.sum()).backward()


def _worker(rank, world, port, n_views, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sc = small_scene(n=60, w=32, h=32, views=n_views, seed=3, f=4)
    p = _params(sc)
    grads = FlatGrads(p.values())
    sharded_step(lambda v: _render_view(sc, p, v), n_views, grads, rank, world)
    # every rank holds the same, fully reduced buffer
    gathered = [torch.zeros_like(grads.flat) for _ in range(world)]
    dist.all_gather(gathered, grads.flat)
    assert all(torch.equal(gathered[0], t) for t in gathered)
    assert p["means"].grad.data_ptr() == grads.flat.data_ptr() + grads.offsets[0] * 8
    if rank == 0:
        ret.put({k: v.grad.numpy().copy() for k, v in p.items()})
    dist.barrier()
    dist.destroy_process_group()


def _exchange_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    shapes = dict(big_a=(300, 16, 3), small_a=(300, 3), small_b=(300,), big_b=(300, 32), unused=(300, 4))
    base = {k: torch.randn(s, generator=g, dtype=torch.float64) for k, s in shapes.items()}
    params = {k: v.clone().requires_grad_(True) for k, v in base.items()}
    ex = GradExchange(params.values(), big_bytes=300 * 32 * 8)     # big_a and big_b reduce in place
    assert [p.shape for p in ex.big] == [params["big_a"].shape, params["big_b"].shape]
    for k, p in params.items():
        if k == "unused" and rank == 1:
            continue                                                # this rank never reaches the parameter
        p.grad = base[k] * (rank + 1)                               # rank r contributes (r+1) * base
    params["small_a"].grad = params["small_a"].grad.t().contiguous().t()   # a non-contiguous gradient
    ex.all_reduce_mean()
    if rank == 0:
        ret.put({k: p.grad.numpy().copy() for k, p in params.items()})   # plain arrays: no shared-memory handles
        #                                                             that die with this process
    dist.barrier()
    dist.destroy_process_group()


def test_in_place_gradient_exchange_averages_over_ranks():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    got = ret.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(5)
    shapes = dict(big_a=(300, 16, 3), small_a=(300, 3), small_b=(300,), big_b=(300, 32), unused=(300, 4))
    base = {k: torch.randn(s, generator=g, dtype=torch.float64) for k, s in shapes.items()}
    for k, v in got.items():
        want = base[k] * (0.5 if k == "unused" else 1.5)            # mean of 1x and 2x (or of 1x and nothing)
        assert torch.allclose(torch.from_numpy(v), want, atol=1e-12), k


@pytest.mark.parametrize("n_views", [2, 3])
def test_sharded_step_matches_single_process_mean(n_views):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_views, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=240)
        assert pr.exitcode == 0
    sc = small_scene(n=60, w=32, h=32, views=n_views, seed=3, f=4)
    p = _params(sc)
    for v in range(n_views):
        _render_view(sc, p, v)
    for k in p:
        ref = p[k].grad / n_views
        assert torch.allclose(torch.from_numpy(got[k]), ref, rtol=1e-10, atol=1e-12), k


def test_view_assignment_covers_every_view_once():
    for world in (1, 2, 4, 8):
        for n_views in (1, 3, 8, 13):
            seen = sorted(v for r in range(world) for v in view_for_rank(n_views, r, world))
            assert seen == list(range(n_views))


def test_flat_grads_views_alias_one_buffer():
    a = torch.zeros(5, 3, requires_grad=True)
    b = torch.zeros(7, requires_grad=True)
    fg = FlatGrads([a, b])
    (a.sum() * 2 + (b * 3).sum()).backward()
    assert fg.flat[:15].eq(2).all() and fg.flat[fg.offsets[1]:fg.offsets[1] + 7].eq(3).all()
    fg.zero_()
    assert a.grad.abs().sum() == 0 and a.grad.data_ptr() == fg.flat.data_ptr()
    a.grad = None
    fg.zero_()
    assert a.grad is not None and a.grad.data_ptr() == fg.flat.data_ptr()


# ---------------------------------------------------------------------------------------------------------------
# sharded exchange: reduce-scatter -> optimiser shard -> all-gather, and refinement with exchanged statistics
# ---------------------------------------------------------------------------------------------------------------
class _TorchAdamShards:
    """The ``update`` callback of ShardedExchange with torch CPU operators (float64): Adam on the slices."""

    def __init__(self, lrs):
        self.lrs, self.state = lrs, {}

    def __call__(self, segments):
        for name, p, g, lo in segments:
            st = self.state.setdefault((name, lo, p.numel()), {"step": 0, "exp_avg": torch.zeros_like(p),
                                                               "exp_avg_sq": torch.zeros_like(p)})
            st["step"] += 1
            st["exp_avg"].mul_(0.9).add_(g, alpha=0.1)
            st["exp_avg_sq"].mul_(0.999).addcmul_(g, g, value=0.001)
            bc1, bc2 = 1 - 0.9 ** st["step"], 1 - 0.999 ** st["step"]
            denom = (st["exp_avg_sq"].sqrt() / bc2 ** 0.5).add_(1e-15)
            p.addcdiv_(st["exp_avg"], denom, value=-self.lrs[name] / bc1)

    # same interface as pogs_amd.dist.ShardAdam for the refinement round trip
    from pogs_amd.dist import ShardAdam as _SA
    full_state = _SA.full_state
    load_full_state = _SA.load_full_state


_LRS = {"means": 1e-2, "big": 3e-2, "opac": 5e-2, "unused": 1e-2}


def _toy_params(n):
    g = torch.Generator().manual_seed(11)
    return {"means": torch.randn(n, 3, generator=g, dtype=torch.float64),
            "big": torch.randn(n, 16, 3, generator=g, dtype=torch.float64),
            "opac": torch.randn(n, generator=g, dtype=torch.float64),
            "unused": torch.randn(n, 4, generator=g, dtype=torch.float64)}


def _toy_loss(p, rank, step):
    """A per-'view' loss: different on every rank and step; ``unused`` is reached by rank 0 only."""
    w = 1.0 + rank + 0.1 * step
    loss = (p["means"] ** 2).sum() * w + (p["big"].sin() * w).sum() + (p["opac"] * p["means"][:, 0]).sum() * (2 - rank)
    if rank == 0:
        loss = loss + (p["unused"] ** 3).sum()
    return loss


def _refine_identically(p, moments, stats, gen):
    """Stand-in for a densification step: prune where the exchanged statistic is small, duplicate where it is large,
    perturb the copies with samples from the shared generator.  Returns new params and moments (same on all ranks)."""
    keep = stats >= stats.median() * 0.5
    dup = stats > stats.median() * 1.5
    idx = torch.cat([torch.nonzero(keep).flatten(), torch.nonzero(dup).flatten()])
    noise = torch.randn(int(dup.sum()), 3, generator=gen, dtype=torch.float64) * 1e-3
    new_p = {k: v.detach()[idx].clone() for k, v in p.items()}
    new_p["means"][int(keep.sum()):] += noise
    new_m = {k: {"step": m["step"], "exp_avg": m["exp_avg"][idx].clone(), "exp_avg_sq": m["exp_avg_sq"][idx].clone()}
             for k, m in moments.items()}
    for m in new_m.values():                       # moments of new Gaussians start at zero
        m["exp_avg"][int(keep.sum()):] = 0
        m["exp_avg_sq"][int(keep.sum()):] = 0
    return new_p, new_m


def _sharded_worker(rank, world, port, ret, direct):
    from pogs_amd.dist import ShardedExchange, sync_strategy_state, wait_ready

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    n = 301                                                   # odd sizes: shards cut through parameters, padding
    p = {k: v.clone().requires_grad_(True) for k, v in _toy_params(n).items()}
    opt = _TorchAdamShards(_LRS)
    ex = ShardedExchange(p, opt, big_bytes=n * 48 * 8, direct=direct)     # "big" alone, the rest share a bucket
    assert [b.names for b in ex.buckets] == [["big"], ["means", "opac", "unused"]]
    stats = {"grad2d": torch.zeros(n, dtype=torch.float64), "count": torch.zeros(n, dtype=torch.float64),
             "radii": torch.zeros(n, dtype=torch.float64)}
    gen = torch.Generator().manual_seed(77)                   # shared seed: every rank draws the same samples
    for step in range(4):
        for t in p.values():
            t.grad = None
        ex.begin_step(wait=["means"])
        for t in p.values():
            wait_ready(t)
        _toy_loss(p, rank, step).backward()
        stats["grad2d"] += p["means"].grad.norm(dim=-1).detach() * (1 + rank)     # per-rank statistics
        stats["count"] += 1
        stats["radii"] = torch.maximum(stats["radii"], p["opac"].detach().abs() * (1 + rank))
        ex.finish_step()
        if step == 1:                                         # a refinement between steps 1 and 2
            ex.wait_all()
            sync_strategy_state(stats)
            moments = opt.full_state(ex)
            new_p, new_m = _refine_identically({k: v for k, v in p.items()}, moments,
                                               stats["grad2d"] / stats["count"] + stats["radii"], gen)
            p = {k: v.requires_grad_(True) for k, v in new_p.items()}
            ex.rebuild(p)
            opt.load_full_state(ex, new_m)
            stats = {k: torch.zeros(len(p["means"]), dtype=torch.float64) for k in stats}
    ex.wait_all()
    flat = torch.cat([v.detach().reshape(-1) for v in p.values()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    assert all(torch.equal(gathered[0], t) for t in gathered), "ranks hold different parameters"
    if rank == 0:
        ret.put({k: v.detach().numpy().copy() for k, v in p.items()})
    dist.barrier()
    dist.destroy_process_group()


def _sharded_reference(world=2):
    """The same four steps in one process: mean of the per-rank gradients, whole-tensor Adam, same refinement."""
    n = 301
    p = {k: v.clone().requires_grad_(True) for k, v in _toy_params(n).items()}
    opts = {k: torch.optim.Adam([v], lr=_LRS[k], eps=1e-15) for k, v in p.items()}
    stats = {"grad2d": torch.zeros(n, dtype=torch.float64), "count": torch.zeros(n, dtype=torch.float64),
             "radii": torch.zeros(n, dtype=torch.float64)}
    gen = torch.Generator().manual_seed(77)
    for step in range(4):
        grads = {k: torch.zeros_like(v) for k, v in p.items()}
        for rank in range(world):
            for t in p.values():
                t.grad = None
            _toy_loss(p, rank, step).backward()
            for k, v in p.items():
                if v.grad is not None:
                    grads[k] += v.grad / world
            stats["grad2d"] += p["means"].grad.norm(dim=-1).detach() * (1 + rank)
            stats["count"] += 1
            stats["radii"] = torch.maximum(stats["radii"], p["opac"].detach().abs() * (1 + rank))
        for k, v in p.items():
            v.grad = grads[k]
            opts[k].step()
        if step == 1:
            moments = {k: {"step": int(opts[k].state[v]["step"]), "exp_avg": opts[k].state[v]["exp_avg"],
                           "exp_avg_sq": opts[k].state[v]["exp_avg_sq"]} for k, v in p.items()}
            new_p, new_m = _refine_identically(p, moments, stats["grad2d"] / stats["count"] + stats["radii"], gen)
            p = {k: v.requires_grad_(True) for k, v in new_p.items()}
            opts = {k: torch.optim.Adam([v], lr=_LRS[k], eps=1e-15) for k, v in p.items()}
            for k, v in p.items():
                opts[k].state[v] = {"step": torch.tensor(float(new_m[k]["step"])), "exp_avg": new_m[k]["exp_avg"],
                                    "exp_avg_sq": new_m[k]["exp_avg_sq"]}
            stats = {k: torch.zeros(len(p["means"]), dtype=torch.float64) for k in stats}
    return {k: v.detach() for k, v in p.items()}


@pytest.mark.parametrize("direct", [True, False])
def test_sharded_exchange_matches_mean_gradient_adam_through_a_refinement(direct):
    """world 2, gloo: reduce-scatter (direct form: all-to-all + local sum; or the library form's bookkeeping -- in-place
    shard view of the gradient buffer, bucket padding, mean -- with an all-reduce standing in for the reduce-scatter
    gloo lacks) -> Adam on the local shard -> all-gather of the parameters for
    four steps with a refinement (exchanged statistics, shared random samples, rebuilt layout, re-sharded moments)
    after the second; both ranks end with the same parameters, equal to single-process mean-gradient Adam."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, ret, direct)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = ret.get(timeout=240)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    want = _sharded_reference()
    for k, v in want.items():
        assert got[k].shape == tuple(v.shape), k
        assert torch.allclose(torch.from_numpy(got[k]), v, rtol=1e-9, atol=1e-12), k


def test_sharded_exchange_single_process_is_plain_adam():
    from pogs_amd.dist import ShardedExchange

    p = {k: v.clone().requires_grad_(True) for k, v in _toy_params(50).items()}
    q = {k: v.clone().requires_grad_(True) for k, v in _toy_params(50).items()}
    ex = ShardedExchange(p, _TorchAdamShards(_LRS), big_bytes=1 << 30)
    opts = {k: torch.optim.Adam([v], lr=_LRS[k], eps=1e-15) for k, v in q.items()}
    for step in range(3):
        for t in list(p.values()) + list(q.values()):
            t.grad = None
        ex.begin_step()
        _toy_loss(p, 0, step).backward()
        ex.finish_step()
        _toy_loss(q, 0, step).backward()
        for o in opts.values():
            o.step()
    for k in p:
        assert torch.allclose(p[k], q[k], rtol=1e-12, atol=1e-14), k
