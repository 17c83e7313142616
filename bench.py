#!/usr/bin/env python3
"""Headline benchmark: Mpix/s of one rasterization forward+backward at the POGS training-step shape
(1M Gaussians, 1920x1080, SH-RGB + expected depth + 32 feature channels) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W [--config cfg2|cfg1|cfg4|cfg5|cfg3]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one `rasterization()` call forward + backward (projection, SH, binning + sort, compositing and
every backward kernel) on one camera per GPU with inputs resident in HBM.  Every step renders another camera of a
ring around the scene, as the reference's trainer draws another camera each iteration.  With N > 1 every rank
renders a different view of the same (replicated) scene and the parameter gradients are exchanged over RCCL
inside the step (weak scaling: per-GPU work fixed).  Rank 0 prints ONE JSON line.

`--config` selects another of the workload shapes (cfg1 plumbing shape, cfg5 4K stress shape, cfg3 the tracker's
50-iteration pose loop); the default, cfg2, is the headline.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pogs_amd import _lib, rendering  # noqa: E402
from pogs_amd.dist import make_exchange  # noqa: E402
from pogs_amd.scenes import make_scene  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E nominal
FP32_VECTOR_PEAK_TFLOPS = 157.3

CONFIGS = {
    # name: (gaussians, width, height, feature channels, tile size)
    "cfg1": (10_000, 256, 256, 16, 16),
    "cfg2": (1_000_000, 1920, 1080, 32, 16),
    "cfg4": (3_000_000, 1920, 1080, 32, 16),     # multi-view training: 8 ring cameras, one view per GPU and step
    "cfg5": (6_000_000, 3840, 2160, 64, 16),
}


def stage_bytes(N, C, I, P, K, F, D, key_bits):
    """Algorithmic bytes per entry point for one forward+backward call over I (Gaussian, tile) pairs; they sum to
    N(212+36K+8F) + I(108+24s+12D) + P(8D+20) with s = ceil(key_bits/8) radix passes."""
    s = math.ceil(key_bits / 8)
    cn = C * N
    b = {
        "pr_project_fwd": cn * (44 + 28),
        "pr_sh_fwd": cn * (12 * K + 12),
        "pr_isect_count": 0,
        # the two-level sort is priced as the reference's one sort: its bytes are charged to the second level
        "pr_depth_order": 0,
        "pr_isect_emit_sort": I * (12 + 24 * s) + I * 8,   # incl. the offsets table
        "pr_raster_fwd": I * (28 + 4 * D) + P * (4 * D + 8),
        "pr_raster_bwd": I * (28 + 4 * D) + I * (32 + 4 * D) + P * (4 * D + 12),
        "pr_project_bwd": cn * (28 + 44 + 44),
        "pr_sh_bwd": cn * (12 + 12 * K + 12 * K),
    }
    total = cn * (212 + 36 * K + 8 * F) + I * (108 + 24 * s + 12 * D) + P * (8 * D + 20)
    b["other"] = total - sum(b.values())  # feature read/write + gradient hand-offs between entry points
    return b, total


def cpu_baseline(sc, F, tile_size, target_s=25.0):
    """The oracle (a CPU port of the algorithm) timed on this host's cores on a bounded sample of the
    same workload: full projection + SH + binning for all Gaussians, compositing forward+backward on an
    evenly spread subset of tiles sized to ~target_s seconds."""
    from oracle import gsplat_ref as R
    from oracle import raster_c as RC

    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    W, H = sc.width, sc.height
    t0 = time.perf_counter()
    with torch.no_grad():
        scales, opac = sc.log_scales.exp(), torch.sigmoid(sc.opacity_logits)
        radii, means2d, depths, conics, _ = R.fully_fused_projection(sc.means, sc.quats, scales, sc.viewmats[:1],
                                                                     sc.Ks[:1], W, H)
        dirs = sc.means[None] - torch.inverse(sc.viewmats[:1])[:, None, :3, 3]
        rgb = torch.clamp_min(R.spherical_harmonics(3, dirs, sc.sh[None], masks=radii > 0) + 0.5, 0.0)
        cols = torch.cat([rgb, sc.features[None], depths[..., None]], dim=-1).contiguous()
        tw, th = math.ceil(W / tile_size), math.ceil(H / tile_size)
        _, ids, flat = R.isect_tiles(means2d, radii, depths, tile_size, tw, th)
        offs = R.isect_offset_encode(ids, 1, tw, th)
    t_front = time.perf_counter() - t0
    D = cols.shape[-1]
    opac = opac[None].contiguous()
    g = torch.Generator().manual_seed(1)

    def run(stride):
        tiles = torch.arange(0, tw * th, stride, dtype=torch.int32)
        t1 = time.perf_counter()
        out, alpha, last = RC.raster_fwd(means2d, conics, cols, opac, W, H, tile_size, offs, flat, tiles=tiles)
        vc = torch.randn(out.shape, generator=g)
        va = torch.randn(alpha.shape, generator=g)
        t2 = time.perf_counter()
        RC.raster_bwd(means2d, conics, cols, opac, W, H, tile_size, offs, flat, alpha, last, vc, va, tiles=tiles)
        t3 = time.perf_counter()
        return tiles.numel(), (t2 - t1) - 0.0 + (t3 - t2)

    # probe on a quarter of the tiles (enough of them to load every core evenly), then size the sample
    n_probe, t_probe = run(4)
    per_tile = max(t_probe / n_probe, 1e-6)
    stride = max(1, int(math.ceil(tw * th * per_tile / target_s)))
    reps = max(1, int(target_s / max(tw * th * per_tile / stride, 1e-3) / 1.5))  # fast hosts: repeat the pass
    n_tiles, t_rast = 0, 0.0
    for _ in range(reps):
        n, t = run(stride)
        n_tiles, t_rast = n_tiles + n, t_rast + t
    frac = n_tiles / float(tw * th)          # frames' worth of tiles composited
    pixels = frac * W * H
    t_total = t_rast + t_front * frac
    return {
        "value": pixels / t_total / 1e6, "unit": "Mpix/s", "cores": cores, "kind": "port",
        "sample": (f"{sc.n}-Gaussian {W}x{H} scene, view 0: projection+SH+binning of all Gaussians in PyTorch CPU "
                   f"({t_front:.1f} s per frame, charged pro rata), C/OpenMP compositing fwd+bwd of every "
                   f"{stride}th tile, {reps} pass(es) ({n_tiles} tiles = {frac:.2f} frames, {t_rast:.1f} s), D={D}; "
                   f"projection/SH backward not included"),
    }


def copy_peak_gbs(dev):
    """Measured HBM peak beside the nominal one: a 1 GiB device-to-device copy (read + write)."""
    a_buf = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b_buf = torch.empty_like(a_buf)
    for _ in range(2):
        b_buf.copy_(a_buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b_buf.copy_(a_buf)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * a_buf.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pmc_record(workload_key):
    """Per-launch HBM bytes of the entry points from this build's counter runs (profiles/pmc_traffic.json, written
    by scripts/pmc_summary.py), when they were taken on this workload."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(pmc))
    except Exception:
        return {}
    return rec if rec.get("workload") == workload_key else {}


def pair_counts(workload_key, pixels_per_lane):
    """Iteration counts of the backward compositor from a diagnostic build, for the kernel variant that ran
    (profiles/r03_bwd_pair_counts.json: 8x8 regions; profiles/r02_bwd_pair_counts.json: half tiles)."""
    for name in ("r03_bwd_pair_counts.json", "r02_bwd_pair_counts.json"):
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if rec.get("workload") == workload_key and rec.get("pixels_per_lane", 4) == pixels_per_lane:
            rec["file"] = "profiles/" + name
            return rec
    return None


def tracking_config(args, dev):
    """cfg3: the tracker's per-frame pose optimisation (pogs/tracking/rigid_group_optimizer.py:402-453), 50
    iterations per frame on 200k Gaussians at 640x360.  Single GPU in the reference: replicas only."""
    from pogs_amd.tracking import TrackingLoop

    loop = TrackingLoop.synthetic(dev, n=200_000, width=640, height=360, groups=4, seed=0)
    iters = 50

    first = []

    def timed():
        for _ in range(max(args.warmup, 1)):
            out = loop.frame(iters)
            if not first:
                first.append(out[0])          # the loss before any pose update of this run
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for _ in range(args.steps):
            out = loop.frame(iters)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3, out

    loop.fused_transform = False        # the reference's structure: the transform as its own pass over the model
    ms_two_step, _ = timed()
    loop.fused_transform = True         # default: the transform inside the projection kernels
    loop.share_binning = False          # tile 8 then tile 16, a binning each (only the depth order shared)
    ms_two_binnings, _ = timed()
    loop.fuse_renders = True            # RGB+ED and the features as one 68-channel render
    ms_fused, _ = timed()
    loop.fuse_renders = False
    loop.share_binning = True           # default: both renders at tile 16 over one binning
    loop.skip_unused_rgb = True         # the first render as "ED": no loss reads the colours
    ms_no_rgb, _ = timed()
    loop.skip_unused_rgb = False
    ms, losses = timed()

    # every entry point's share of an iteration (HIP events, separate pass), and the dominant compositor launch
    full = rendering.StageTimer()
    rendering.set_stage_timer(full)
    loop.frame(iters)
    torch.cuda.synchronize()
    rendering.set_stage_timer(None)
    stages = {k: v[1] / iters for k, v in full.summary().items()}            # ms per iteration, both renders together
    with torch.no_grad():
        common = dict(viewmats=loop.viewmat, Ks=loop.K, width=loop.width, height=loop.height, packed=False,
                      rigid=(loop.labels, loop.init_p2w, loop.part_deltas.detach()))
        timer = rendering.StageTimer(only=("pr_raster_fwd",))
        rendering.set_stage_timer(timer)
        for _ in range(10):
            _, _, meta = rendering.rasterization(loop.means, loop.quats, loop.scales, loop.opacities, loop.features,
                                                 **common)
        torch.cuda.synchronize()
        rendering.set_stage_timer(None)
    n_calls, t_ms = timer.summary()["pr_raster_fwd"]
    I, P, D = meta["n_composited_isects"], loop.width * loop.height, 64
    dom_bytes = I * (28 + 4 * D) + P * (4 * D + 8)
    avg_ms = t_ms / n_calls
    achieved = dom_bytes / (avg_ms * 1e-3) / 1e9

    # the per-object branch (the reference tracker's default): one crop camera per rigid group, all crops of a pass in
    # one render call (rasterization(camera_groups=...)); and as the reference structures it, one call per object
    roi_loop = TrackingLoop.synthetic(dev, n=200_000, width=640, height=360, groups=4, seed=0, rois=True)
    for _ in range(max(args.warmup, 1)):
        roi_loop.frame(iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        roi_loop.frame(iters)
    torch.cuda.synchronize()
    ms_roi = (time.perf_counter() - t0) / args.steps * 1e3
    roi_loop.batch_rois = False
    roi_loop.frame(iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(args.steps // 4, 1)):
        roi_loop.frame(iters)
    torch.cuda.synchronize()
    ms_roi_loop = (time.perf_counter() - t0) / max(args.steps // 4, 1) * 1e3

    cpu = None
    if not args.no_cpu_baseline:
        from pogs_amd.scenes import make_scene
        sc = make_scene(200_000, 640, 360, n_features=64, seed=0)
        port = cpu_baseline(sc, 64, 16, target_s=15.0)
        per_iter_ms = P / (port["value"] * 1e6) * 1e3
        cpu = {"value": per_iter_ms * iters, "unit": "ms/frame", "cores": port["cores"], "kind": "port",
               "sample": "50 x one fused RGB+ED+64-feature render forward+backward of the same scene by the oracle "
                         "(stands for an iteration's two renders; MLP, losses, pose update not included): "
                         + port["sample"]}
    return {
        "metric": "ms/frame tracking (200k Gauss, 640x360, 50 iters)", "value": ms, "unit": "ms/frame", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": False,
        "scaling": "replicas only", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "tracking loop: per iteration rigid-group transform, RGB+ED render and 64-channel "
                               "feature render at tile 16 over one binning + channel MLP, fused loss, backward to "
                               "[G,7] pose deltas, Adam",
                   "gaussians": 200_000, "width": 640, "height": 360, "iterations_per_frame": iters, "groups": 4,
                   "loss_first_to_last": [first[0], losses[-1]],       # first iteration of the first frame, last of the last
                   "transform": "inside the projection kernels", "ms_per_frame_transform_as_own_pass": ms_two_step,
                   "ms_per_frame_tile8_then_tile16_a_binning_each": ms_two_binnings,
                   "ms_per_frame_one_68_channel_render": ms_fused,
                   "ms_per_frame_first_render_depth_only": ms_no_rgb,
                   "ms_per_frame_per_object_crops": ms_roi, "ms_per_frame_per_object_crops_one_call_per_object": ms_roi_loop,
                   "crop_sizes": [(r["width"], r["height"], int(r["ids"].numel())) for r in roi_loop.rois],
                   "stages_ms_per_iteration": {k: round(v, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])}},
        "roofline": {"bound": "hbm", "kernel": "pr_raster_fwd of the 64-channel feature render", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": avg_ms, "n_isects_composited": I,
                     "limited_by": "920 tiles: the launch is a few long dependent chains per SIMD, not bandwidth"},
        "cpu_baseline": cpu,
    }


def pogs_step_config(args, dev):
    """The training iteration POGS runs once its feature fields are active: three rasterization() calls on one scene
    and one camera (pogs/pogs.py:352-371 RGB+ED at full resolution with absgrad; :419-437 the 96-channel hash-grid
    encoding at half resolution, geometry and colour gradients, absgrad; :493-510 the 64 DINO channels at 450x250
    with every geometry input detached), each followed by its backward pass.  The hash-grid output and the losses
    are stood in for by a random [N,96] leaf and random cotangents."""
    N, W, H, ts = args.gaussians or 1_000_000, args.width or 1920, args.height or 1080, args.tile_size or 16
    n_views = max(args.views, 1)
    sc = make_scene(N, W, H, n_features=64, n_views=n_views, seed=0)
    g = torch.Generator().manual_seed(7)
    params = {"means": sc.means, "quats": sc.quats, "log_scales": sc.log_scales, "opacity_logits": sc.opacity_logits,
              "sh": sc.sh, "dino": sc.features, "clip": torch.randn(N, 96, generator=g) * 0.5}
    params = {k: v.to(dev).contiguous().requires_grad_(True) for k, v in params.items()}
    viewmats, Ks = sc.viewmats.to(dev).contiguous(), sc.Ks.to(dev).contiguous()
    W2, H2, W3, H3 = W // 2, H // 2, 450 * W // 1920, 250 * H // 1080
    Ks2, Ks3 = Ks.clone(), Ks.clone()
    Ks2[:, :2, :] *= 0.5
    Ks3[:, :2, :] *= W3 / W
    calls = [
        dict(name="rgb_ed", channels=4, width=W, height=H, geometry=True, what="SH-RGB(deg 3)+ED, absgrad"),
        dict(name="clip_96", channels=96, width=W2, height=H2, geometry=True, what="96 channels, geometry + colour "
             "gradients, absgrad"),
        dict(name="dino_64", channels=64, width=W3, height=H3, geometry=False, what="64 channels, colour gradients only"),
    ]
    cot = [torch.randn(1, c["height"], c["width"], c["channels"], generator=g).to(dev) for c in calls]
    cot_alpha = torch.randn(1, H, W, 1, generator=g).to(dev)
    info = {"counter": 0, "isects": [[], [], []], "share": False, "order": None}

    def render(k, view):
        vm = viewmats[view:view + 1]
        scales, opac = torch.exp(params["log_scales"]), torch.sigmoid(params["opacity_logits"])
        common = dict(viewmats=vm, tile_size=ts, packed=False, near_plane=0.01, far_plane=1e10, sparse_grad=False,
                      rasterize_mode="classic")
        if k > 0 and info["share"]:
            common["depth_order"] = info["order"]     # same positions, same view matrix: the first render's sort
        if k == 0:
            r, a, meta = rendering.rasterization(means=params["means"], quats=params["quats"], scales=scales,
                                                 opacities=opac, colors=params["sh"], sh_degree=3, Ks=Ks[view:view + 1],
                                                 width=W, height=H, render_mode="RGB+ED", absgrad=True, **common)
            torch.autograd.backward([r, a], [cot[0], cot_alpha])
            info["order"] = meta["depth_order"]
        elif k == 1:
            r, a, meta = rendering.rasterization(means=params["means"], quats=params["quats"], scales=scales,
                                                 opacities=opac, colors=params["clip"], Ks=Ks2[view:view + 1],
                                                 width=W2, height=H2, render_mode="RGB", absgrad=True, **common)
            torch.autograd.backward([r], [cot[1]])
        else:
            r, a, meta = rendering.rasterization(means=params["means"].detach(), quats=params["quats"].detach(),
                                                 scales=scales.detach(), opacities=opac.detach(), colors=params["dino"],
                                                 Ks=Ks3[view:view + 1], width=W3, height=H3, render_mode="RGB",
                                                 absgrad=False, **common)
            torch.autograd.backward([r], [cot[2]])
        info["isects"][k].append(meta["n_composited_isects"])

    def step(only=None):
        i = info["counter"]
        info["counter"] = i + 1
        for p in params.values():
            p.grad = None
        for k in range(3):
            if only is None or only == k:
                render(k, i % n_views)

    t_setup, n_setup = time.perf_counter(), 0
    while n_setup < n_views or (time.perf_counter() - t_setup < args.precondition_s and n_setup < 1000):
        step()
        n_setup += 1
        if n_setup % 8 == 0:
            torch.cuda.synchronize()
    info["counter"] = 0
    for _ in range(args.warmup):
        step()
    for lst in info["isects"]:
        lst.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue = (time.perf_counter() - t0) / args.steps       # what the host needs to issue a step
    torch.cuda.synchronize()
    elapsed = (time.perf_counter() - t0) / args.steps
    isects = [sum(lst) / len(lst) for lst in info["isects"]]
    # the same iteration with the first render's depth order handed to the other two (an argument gsplat does not
    # have: the integration adds it to the second and third call)
    info["share"] = True
    for _ in range(max(args.warmup, 2)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed_shared = (time.perf_counter() - t0) / args.steps
    info["share"] = False

    # per call: its own timed pass (wall clock), then a pass with HIP events around every entry point
    per_call = []
    for k, c in enumerate(calls):
        for _ in range(2):
            step(only=k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(only=k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / args.steps * 1e3
        timer = rendering.StageTimer()
        rendering.set_stage_timer(timer)
        for _ in range(args.steps):
            step(only=k)
        torch.cuda.synchronize()
        rendering.set_stage_timer(None)
        stages = {n: v[1] / args.steps for n, v in timer.summary().items()}
        D, n_tiles = c["channels"], math.ceil(c["width"] / ts) * math.ceil(c["height"] / ts)
        plan_f = _lib.raster_plan(D, ts, n_tiles)
        plan_b = _lib.raster_plan(D, ts, n_tiles, backward=True, geom_grad=c["geometry"])
        per_call.append({
            "call": c["name"], "what": c["what"], "width": c["width"], "height": c["height"], "channels": D,
            "ms_alone": ms, "n_isects_composited": isects[k],
            "kernels": {"fwd": {"pixels_per_lane": plan_f[0], "channels_per_pass": plan_f[1], "matrix_cores": plan_f[2]},
                        "bwd": {"pixels_per_lane": plan_b[0], "channels_per_pass": plan_b[1], "matrix_cores": plan_b[2]}},
            "stages_ms": {n: round(v, 4) for n, v in sorted(stages.items(), key=lambda kv: -kv[1])}})
    # roofline of the slowest compositing launch of the step
    k_dom, dom = max(((k, n) for k in range(3) for n in ("pr_raster_fwd", "pr_raster_bwd")),
                     key=lambda kn: per_call[kn[0]]["stages_ms"].get(kn[1], 0.0))
    c = calls[k_dom]
    D, P = c["channels"], c["width"] * c["height"]
    n_tiles = math.ceil(c["width"] / ts) * math.ceil(c["height"] / ts)
    per_stage, _ = stage_bytes(N, 1, isects[k_dom], P, 16, 0, D, 32 + n_tiles.bit_length())
    dom_bytes = per_stage[dom]
    if dom == "pr_raster_bwd" and not c["geometry"]:
        dom_bytes -= isects[k_dom] * 32       # no geometry gradient rows written
    avg_ms = per_call[k_dom]["stages_ms"][dom]
    achieved = dom_bytes / (avg_ms * 1e-3) / 1e9
    pixels = sum(cc["width"] * cc["height"] for cc in calls)
    return {
        "metric": "Mpix/s fwd+bwd over the three renders of a POGS training iteration (1M Gauss; 1080p RGB+ED, "
                  "960x540x96ch, 450x250x64ch)",
        "value": pixels / elapsed / 1e6, "unit": "Mpix/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "pogs_step: %d Gaussians, one camera per step cycling over %d ring cameras, three "
                               "rasterization() calls each followed by its backward pass" % (N, n_views),
                   "gaussians": N, "tile_size": ts, "calls": per_call,
                   "sum_of_calls_alone_ms": sum(pc["ms_alone"] for pc in per_call),
                   "ms_per_step_shared_depth_order": elapsed_shared * 1e3,
                   "host_enqueue_ms_per_step": host_enqueue * 1e3},
        "roofline": {"bound": "hbm", "kernel": "%s of call %s" % (dom, c["name"]), "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": avg_ms,
                     "limited_by": "vector / float32 matrix instruction issue"},
        "cpu_baseline": None,
    }


_RESULT_FD = None


def keep_stdout_for_the_result():
    """Libraries write to file descriptor 1 behind Python's back (RCCL prints a five-line version banner there when
    its first communicator comes up): from here on descriptor 1 is the error stream, and the one JSON line goes to
    the original standard output through ``emit``."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line: str) -> None:
    sys.stdout.flush()
    data = (line + "\n").encode()
    fd = 1 if _RESULT_FD is None else _RESULT_FD
    while data:
        data = data[os.write(fd, data):]


def main():
    keep_stdout_for_the_result()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS) + ["cfg3", "pogs_step"])
    ap.add_argument("--gaussians", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--features", type=int, default=None)
    ap.add_argument("--tile-size", type=int, default=None)
    ap.add_argument("--views", type=int, default=8, help="cameras on the ring the steps cycle through")
    ap.add_argument("--precondition-s", type=float, default=0.5,
                    help="seconds of untimed steps before the warm-up steps (0 = one render per ring camera only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra timed passes (other kernel variants)")
    ap.add_argument("--two-call", action="store_true",
                    help="render RGB+ED and the features as two gsplat-signature calls instead of one fused call")
    ap.add_argument("--with-optimizer", action="store_true",
                    help="N = 1: run the step with the gradient exchange object of N > 1 (no collective with one rank, "
                         "the fused Adam on the whole parameters), like for like with the multi-GPU step")
    ap.add_argument("--lr-scale", type=float, default=0.0,
                    help="factor on the reference's learning rates inside the benchmark (0: the optimiser does all its "
                         "work but the parameters, and with them the intersection counts, stay those of step 0)")
    ap.add_argument("--exchange", default="sharded", choices=["sharded", "allreduce"],
                    help="N > 1: reduce-scatter + all-gather of the gradients in overlapped buckets, or one mean "
                         "all-reduce per large gradient after the backward pass")
    args = ap.parse_args()

    if os.environ.get("POGS_BENCH_HANG_DUMP"):   # debugging aid: dump every thread's stack after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["POGS_BENCH_HANG_DUMP"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # one GPU per rank; POGS_BENCH_SHARE_GPU=1 lets the ranks of a rehearsal share the devices that exist
    n_dev = torch.cuda.device_count()
    if local_rank >= n_dev and os.environ.get("POGS_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"LOCAL_RANK {local_rank} but only {n_dev} device(s)")
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    if args.config == "cfg3":
        if world != 1:
            raise SystemExit("cfg3 (tracking) is single-GPU in the reference: run replicas, not ranks")
        emit(json.dumps(tracking_config(args, dev)))
        return
    if args.config == "pogs_step":
        if world != 1:
            raise SystemExit("pogs_step is measured on one GPU")
        emit(json.dumps(pogs_step_config(args, dev)))
        return
    # POGS_BENCH_FORCE_DIST=1: a single rank takes the N > 1 path (process group, barriers, the exchange's library
    # collectives with one participant) -- a rehearsal of the RCCL calls on a one-GPU box
    multi = world > 1 or os.environ.get("POGS_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("POGS_BENCH_BACKEND", "nccl")   # "gloo" for a rehearsal on a shared GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    N, W, H, F, ts = CONFIGS[args.config]
    N, W, H = args.gaussians or N, args.width or W, args.height or H
    F, ts = (F if args.features is None else args.features), args.tile_size or ts
    n_views = max(world, args.views, 1)
    sc = make_scene(N, W, H, n_features=F, n_views=n_views, seed=0)
    params = {
        "means": sc.means, "quats": sc.quats, "log_scales": sc.log_scales, "opacity_logits": sc.opacity_logits,
        "sh": sc.sh, "features": sc.features,
    }
    params = {k: v.to(dev).contiguous().requires_grad_(True) for k, v in params.items()}
    viewmats = sc.viewmats.to(dev).contiguous()
    Ks = sc.Ks.to(dev).contiguous()
    # Multi-GPU: the gradients are exchanged every step (mean over ranks).  Single GPU: plain
    # zero_grad(set_to_none=True) semantics, as the reference's trainer does between iterations.
    # --with-optimizer: the same object with one rank (its collectives drop out, the fused Adam runs on everything).
    ex = {"grads": make_exchange(args.exchange, params, lr_scale=args.lr_scale)
          if multi or args.with_optimizer else None}
    D = 3 + F + 1
    g = torch.Generator().manual_seed(100 + rank)
    v_colors = torch.randn(1, H, W, D, generator=g).to(dev)
    v_alpha = torch.randn(1, H, W, 1, generator=g).to(dev)
    info = {"n_walked": [], "counter": 0}

    def step(two_call=args.two_call):
        i = info["counter"]
        info["counter"] = i + 1
        view = (rank + i) % n_views
        viewmat, K = viewmats[view:view + 1], Ks[view:view + 1]
        for p in params.values():   # autograd adopts the gradient tensors the kernels produce
            p.grad = None
        grads = ex["grads"]
        if grads is not None:   # the forward waits for each parameter of the previous exchange where it first reads it
            grads.begin_step(wait=["means", "quats", "log_scales", "opacity_logits"])
        scales = torch.exp(params["log_scales"])
        opac = torch.sigmoid(params["opacity_logits"])
        common = dict(means=params["means"], quats=params["quats"], scales=scales, opacities=opac, viewmats=viewmat,
                      Ks=K, width=W, height=H, tile_size=ts, packed=False, near_plane=0.01, far_plane=1e10,
                      sparse_grad=False, rasterize_mode="classic")
        if two_call:
            r1, a1, meta = rendering.rasterization(colors=params["sh"], sh_degree=3, render_mode="RGB+ED",
                                                   absgrad=True, **common)
            r2, a2, _ = rendering.rasterization(colors=params["features"], sh_degree=None, render_mode="RGB",
                                                absgrad=False, **common)
            torch.autograd.backward([r1, a1, r2, a2],
                                    [v_colors[..., [0, 1, 2, D - 1]], v_alpha, v_colors[..., 3:D - 1], v_alpha])
        else:
            r, a, meta = rendering.rasterization(colors=params["sh"], sh_degree=3, render_mode="RGB+ED",
                                                 absgrad=True, features=params["features"], **common)
            torch.autograd.backward([r, a], [v_colors, v_alpha])
        info["meta"] = meta
        info["n_walked"].append(meta["n_composited_isects"])
        if grads is not None:
            grads.finish_step()

    def timed(n_steps, **kw):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(**kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n_steps

    # set-up, before the W warm-up steps: render every ring camera (sizes the allocator's pools for every view) and keep
    # going for about half a second so that clocks, caches and the intersection-count history are in steady state
    t_setup = time.perf_counter()
    n_setup = 0
    # (with more than one rank every step is a collective: the count must not depend on a local clock)
    fixed = max(n_views, 48) if multi else None
    while (n_setup < fixed) if fixed is not None else (
            n_setup < n_views or (time.perf_counter() - t_setup < args.precondition_s and n_setup < 1000)):
        step()
        n_setup += 1
        if n_setup % 8 == 0:
            torch.cuda.synchronize()
    info["counter"] = 0
    for _ in range(args.warmup):
        step()
    info["n_walked"].clear()
    # HIP events around the two compositing entry points only inside the timed region (the roofline's kernel time);
    # the table of every stage comes from a separate pass afterwards: two events per call on ten calls per step cost
    # about 2 % of the step
    timer = rendering.StageTimer(only=("pr_raster_bwd", "pr_raster_fwd"))
    rendering.set_stage_timer(timer)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rendering.set_stage_timer(None)
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the table of every stage: a separate pass after the timed region -- run by EVERY rank (with more than one rank a
    # step contains collectives), only rank 0 records
    full = rendering.StageTimer()
    if rank == 0:
        rendering.set_stage_timer(full)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    rendering.set_stage_timer(None)
    grads = ex["grads"]
    if grads is not None:
        grads.wait_all()
        torch.cuda.synchronize()
    if multi:
        dist.barrier()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        pixels = world * H * W
        value = pixels / (elapsed / args.steps) / 1e6
        n_tiles = math.ceil(W / ts) * math.ceil(H / ts)
        walked = info["n_walked"]
        I_walked = sum(walked) / len(walked)          # pairs the compositor's lists hold, mean over the timed views
        # gsplat's binning rule (every tile under the 3-sigma box) for the last view -- read here, after the timed
        # region, from the lazily built reference tables; the compositor walks the shorter footprint-tested lists.
        I_gsplat = info["meta"]["flatten_ids"].numel()
        key_bits = 32 + n_tiles.bit_length()
        per_stage, total_bytes = stage_bytes(N, 1, I_walked, H * W, 16, F, D, key_bits)
        per_stage_gsplat, total_gsplat = stage_bytes(N, 1, I_gsplat, H * W, 16, F, D, key_bits)
        summ = timer.summary()
        stages = {k: v[1] / max(v[0], 1) * (v[0] / args.steps) for k, v in full.summary().items()}  # ms per step
        stages.update({k: v[1] / max(v[0], 1) * (v[0] / args.steps) for k, v in summ.items()})
        # the dominant kernel: the slower of the two compositing entry points (the ones timed inside the timed region)
        dom = max((k for k in summ if per_stage.get(k, 0) > 0), key=stages.get)
        launches, total_ms = summ[dom]
        avg_ms = total_ms / launches
        dom_bytes = per_stage[dom] / (launches / args.steps)
        achieved = dom_bytes / (avg_ms * 1e-3) / 1e9
        pmc = pmc_record(f"{N}x{W}x{H}xF{F}ts{ts}")
        # Secondary line: FP32 work of the dominant kernel.  Per (pixel, Gaussian) pair the backward compositor does two
        # channel contractions of D fused multiply-adds each plus ~30 for alpha, transmittance and the geometry
        # gradients.  Which pairs it evaluates is measured, not assumed: the diagnostic build (scripts/diag_bwd.py)
        # counts the (tile, Gaussian) iterations that pass the kernel's "does any pixel take this Gaussian" test and
        # how many of them run over half the tile only; every such iteration evaluates all pixels of the (half) tile.
        fp32 = None
        fwd_plan = _lib.raster_plan(D, ts, n_tiles)
        bwd_plan = _lib.raster_plan(D, ts, n_tiles, backward=True)
        diag = pair_counts(f"{N}x{W}x{H}xF{F}ts{ts}", bwd_plan[0])
        if dom == "pr_raster_bwd" and diag:
            iters, half = diag["contributing_iters"], diag["half_tile_iters"]
            # pixels one wavefront covers: an 8x8 region for the matrix-core kernel, else ts*ts*ppl/4
            region = diag.get("pixels_per_iteration", ts * ts * diag.get("pixels_per_lane", 4) // 4)
            pairs = (iters - half) * region + half * region / 2
            fma = 2 * D + 30
            flops = 2.0 * fma * pairs
            fp32 = {"fma_per_pixel_pair": fma, "region_gaussian_iterations": iters, "of_them_half_tile": half,
                    "pixels_per_iteration": region, "pixel_pairs_evaluated": pairs,
                    "achieved_tflops": flops / (avg_ms * 1e-3) / 1e12, "peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
                    "frac": flops / (avg_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                    "note": "the two channel contractions run as v_mfma_f32_16x16x4_f32, whose rate equals the packed "
                            "vector rate (measured: matrix and vector instructions of one SIMD do not overlap)"
                            if bwd_plan[2] else "all on the vector unit",
                    "source": diag["file"] + " (diagnostic build, view 0)"}
            if "active_px_per_iter" in diag:
                fp32["pixel_pairs_contributing"] = diag["active_px_per_iter"] * iters
        out = {
            "metric": ("Mpix/s fwd+bwd (1M Gauss, 1080p, RGB+32ch feat)" if args.config == "cfg2" and not (
                args.gaussians or args.width or args.height or args.features is not None or args.tile_size)
                else "Mpix/s fwd+bwd (%s: %d Gauss, %dx%d, RGB+ED+%dch feat)" % (args.config, N, W, H, F)),
            "value": value, "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not fwd_plan[2] else "f32 (forward colour accumulation split-bf16 x3 on matrix cores)",
            "data": "synthetic",
            "config": {
                "workload": ("%s: POGS training step shape, %d Gaussians, %dx%d, SH-RGB(deg 3)+ED+%d feature channels, "
                             "fwd+bwd, absgrad, tile %d, 1 view per GPU per step cycling over %d ring cameras%s" % (
                                 args.config, N, W, H, F, ts, n_views,
                                 ", fused single call" if not args.two_call else ", two calls")),
                "gaussians": N, "width": W, "height": H, "channels": D,
                "n_isects_composited": I_walked, "n_isects_composited_per_view": walked[:n_views],
                "n_isects_gsplat_rule_last_view": I_gsplat,
                "kernels": {"fwd": {"pixels_per_lane": fwd_plan[0], "channels_per_pass": fwd_plan[1],
                                    "matrix_cores": fwd_plan[2]},
                            "bwd": {"pixels_per_lane": bwd_plan[0], "channels_per_pass": bwd_plan[1],
                                    "matrix_cores": bwd_plan[2]}},
                "parallelism": ("camera-sharded dp%d, %s" % (world, grads.describe())) if multi else (
                    "single GPU" + (", optimiser step inside the timed region: " + grads.describe()
                                    if grads is not None else "")),
                "optimizer_in_step": grads is not None, "lr_scale": args.lr_scale if grads is not None else None,
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get(dom),
                "limited_by": "vector / float32 matrix instruction issue (see fp32): the kernel moves well under what HBM "
                              "could deliver",
                "measured_copy_peak_GBps": copy_peak_gbs(dev),
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch_gsplat_rule": per_stage_gsplat[dom] / (launches / args.steps),
                "fp32": fp32,
                "whole_call": {"algorithmic_bytes": total_bytes, "achieved": total_bytes / (ms_per_step * 1e-3) / 1e9,
                               "frac": total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "algorithmic_bytes_gsplat_rule": total_gsplat},
            },
            "stages_ms_per_step": {k: round(v, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1])},
            "stages_note": "pr_raster_fwd/bwd: HIP events inside the timed region; the others: a separate pass",
        }
        if not multi and not args.no_variants:
            # the same step with other choices, each timed over the same number of steps after two warm-ups
            variants = {}

            def variant(name, note, **kw):
                for _ in range(2):
                    step(**kw)
                sec = timed(args.steps, **kw)
                variants[name] = {"Mpix/s": H * W / sec / 1e6, "ms_per_step": sec * 1e3, "what": note}

            cur = int(os.environ.get("POGS_RAST_FWD_MFMA", "0") or 0)
            _lib.set_tuning(POGS_RAST_FWD_MFMA=0 if cur else 1)
            variant("fwd_strict_f32" if cur else "fwd_split_bf16",
                    "forward colour accumulation in float32 on the vector ALU (gsplat's arithmetic)" if cur else
                    "forward colour accumulation as three bf16 matrix products (error ~1.2e-5 * max|colour|), opt-in")
            _lib.set_tuning(POGS_RAST_FWD_MFMA=cur if cur else None)
            variant("two_call", "unmodified-POGS call pattern: RGB+ED and the features as two rasterization() calls "
                                "(pogs/pogs.py:352,493)", two_call=not args.two_call)
            _lib.set_tuning(POGS_RAST_BWD_MFMA=0)
            variant("bwd_vector_half_tiles", "backward on the vector unit, one wavefront per 16x8 half tile at 2 pixels "
                    "per lane (raster_bwd_kernel): the default before the float32 matrix-core kernel")
            _lib.set_tuning(POGS_RAST_BWD_MFMA=3)
            variant("bwd_deferred_split_bf16", "backward: colour-gradient contraction of the first 32 channels deferred to "
                    "three bf16 matrix products per 16 contributing Gaussians (raster_bwd_cmfma_kernel), opt-in")
            _lib.set_tuning(POGS_RAST_BWD_MFMA=4, POGS_RAST_BWD_SPLIT=1)
            variant("bwd_split_two_streams", "backward: geometry-only kernel and matrix-core colour-gradient kernel "
                    "(raster_bwd_colour_mfma_kernel) side by side on two streams, opt-in")
            _lib.set_tuning(POGS_RAST_BWD_MFMA=None, POGS_RAST_BWD_SPLIT=None)
            rendering.set_count_speculation(False)
            variant("no_count_speculation", "every call waits on the host for its intersection count, as gsplat does")
            rendering.set_count_speculation(True)
            if ex["grads"] is None and args.exchange == "sharded":
                ex["grads"] = make_exchange(args.exchange, params, lr_scale=args.lr_scale)
                variant("with_optimizer", "the N > 1 step on one GPU: gradient buckets -> fused Adam (learning rates x %g) "
                        "-> parameters, launched from autograd hooks; compare the N > 1 lines with this one" %
                        args.lr_scale)
                ex["grads"].wait_all()
                torch.cuda.synchronize()
                ex["grads"] = None
            out["variants"] = variants
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, F, ts)
        else:
            out["cpu_baseline"] = None
        emit(json.dumps(out))
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:   # a failed collective (or anything else): say so and leave with a failure status
        import traceback
        traceback.print_exc()
        print(json.dumps({"error": "%s: %s" % (type(exc).__name__, exc), "rank": os.environ.get("RANK", "0")}),
              file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(1)     # not sys.exit: a rank stuck in a collective's destructor must not keep the job alive
