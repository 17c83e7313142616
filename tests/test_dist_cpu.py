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
