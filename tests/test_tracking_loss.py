"""The tracker's fused loss (pr_track_loss_fwd / _bwd through pogs_amd.tracking.tracking_loss) against the PyTorch
expression it replaces (pogs/tracking/rigid_group_optimizer.py:364, 373-386), float64 on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(rendered, target, depth, target_depth, valid, scale, cut, mult):
    loss = (target - rendered).norm(dim=-1).nanmean()
    if depth is not None:
        a = torch.clamp(depth / scale, min=1e-8, max=1.0)
        b = torch.clamp(target_depth, min=1e-8, max=1.0)
        if valid is not None:
            a, b = a[valid], b[valid]
        pix = (a - b) ** 2
        pix = pix[pix < cut ** 2]
        if pix.numel() > 0:
            loss = loss + mult * pix.mean()
    return loss


@pytest.mark.parametrize("P,D", [(1, 64), (37, 64), (5000, 64), (333, 4), (100, 128), (230_400, 64)])
@pytest.mark.parametrize("mode", ["features only", "depth", "depth with mask"])
def test_tracking_loss_matches_the_pytorch_expression(gpu, P, D, mode):
    from pogs_amd.tracking import tracking_loss

    g = torch.Generator().manual_seed(P + D)
    rendered = torch.randn(P, D, generator=g, dtype=torch.float64)
    target = torch.randn(P, D, generator=g, dtype=torch.float64)
    if P > 10:
        target[3] = float("nan")            # a pixel the observation has no features for: nanmean skips it
        target[7] = rendered[7]             # distance zero: no gradient
    depth = target_depth = valid = None
    scale, cut, mult = 2.0, 0.26, 3.7
    if mode != "features only":
        depth = torch.rand(P, generator=g, dtype=torch.float64) * 3.0            # some beyond the clamp at scale 2
        target_depth = depth / scale + torch.randn(P, generator=g, dtype=torch.float64) * 0.2   # some beyond the cut
        if mode == "depth with mask":
            valid = torch.rand(P, generator=g) < 0.6
    r64 = rendered.clone().requires_grad_(True)
    d64 = None if depth is None else depth.clone().requires_grad_(True)
    ref = _reference(r64, target, d64, target_depth, valid, scale, cut, mult)
    (ref * 1.7).backward()

    r = rendered.float().to(gpu).requires_grad_(True)
    d = None if depth is None else depth.float().to(gpu).requires_grad_(True)
    got = tracking_loss(r, target.float().to(gpu), d, None if depth is None else target_depth.float().to(gpu),
                        None if valid is None else valid.to(gpu), scale, cut, mult)
    (got * 1.7).backward()
    assert abs(float(got) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref))), (float(got), float(ref))
    # PyTorch's gradient of a row that nanmean skipped is 0 * nan = nan; the kernel gives such a row no gradient
    gr, mine = r64.grad.float(), r.grad.cpu()
    finite = torch.isfinite(gr)
    assert bool((mine[~finite] == 0).all()) and bool(torch.isfinite(mine).all())
    gr = torch.where(finite, gr, torch.zeros_like(gr))
    assert float((mine - gr).abs().max()) <= 1e-5 * float(gr.abs().max()) + 1e-9
    if d is not None:
        # a depth whose squared error sits within float rounding of the cut may fall on either side
        e = (torch.clamp(depth / scale, 1e-8, 1.0) - torch.clamp(target_depth, 1e-8, 1.0)) ** 2
        sure = ((e - cut ** 2).abs() > 1e-6) & (((depth / scale) - 1.0).abs() > 1e-6)
        gd = d64.grad.float()
        assert float(((d.grad.cpu() - gd).abs() * sure).max()) <= 1e-5 * float(gd.abs().max()) + 1e-9


def test_tracking_loss_without_a_finite_pixel_is_nan(gpu):
    from pogs_amd.tracking import tracking_loss

    r = torch.zeros(4, 64, device=gpu)
    t = torch.full((4, 64), float("nan"), device=gpu)
    assert float(tracking_loss(r, t)) != float(tracking_loss(r, t))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_crop_depth_masks_match_the_operator_sequence(gpu, seed):
    """pr_roi_depth_masks against the same masks from PyTorch operators (rigid_group_optimizer.py:316-337: threshold,
    5x5 erosion, rejection above 1.3 x the median, non-zero, 9x9 erosion): crops of different sizes in one padded batch,
    unobserved (NaN) and zero depths, far outliers, a crop without any valid pixel."""
    from pogs_amd.tracking import roi_depth_masks, roi_depth_masks_torch

    gen = torch.Generator().manual_seed(seed)
    sizes = [(70, 98), (56, 42), (84, 112), (28, 14)]
    G, Hm, Wm = len(sizes), 84, 112
    acc = torch.zeros(G, Hm, Wm)
    obs = torch.full((G, Hm, Wm), float("nan"))
    own = torch.zeros(G, Hm, Wm, dtype=torch.bool)
    yy, xx = torch.meshgrid(torch.arange(Hm), torch.arange(Wm), indexing="ij")
    for g, (h, w) in enumerate(sizes):
        own[g, :h, :w] = True
        # a covered blob in the middle with a ragged edge, depth around 0.5 with outliers, holes and zeros
        blob = ((yy - h / 2) / (0.45 * h)) ** 2 + ((xx - w / 2) / (0.45 * w)) ** 2 < 1.0 + 0.2 * torch.rand(Hm, Wm, generator=gen)
        acc[g] = torch.where(blob, 0.86 + 0.14 * torch.rand(Hm, Wm, generator=gen), 0.8 * torch.rand(Hm, Wm, generator=gen))
        d = 0.5 + 0.05 * torch.randn(Hm, Wm, generator=gen)
        r = torch.rand(Hm, Wm, generator=gen)
        d = torch.where(r < 0.002, d * 2.0, d)                              # far behind the object
        d = torch.where((r >= 0.002) & (r < 0.004), torch.zeros_like(d), d)   # zeros
        d = torch.where((r >= 0.004) & (r < 0.006), torch.full_like(d, float("nan")), d)
        obs[g, :h, :w] = d[:h, :w]
    acc[3] = 0.1                                                            # crop 3: nothing covered
    acc, obs, own = acc.to(gpu), obs.to(gpu), own.to(gpu)
    table = torch.tensor(sizes, dtype=torch.int32, device=gpu)
    keep, masked = roi_depth_masks(acc, obs, table)
    keep_ref, masked_ref = roi_depth_masks_torch(acc, obs, table, own)
    assert keep.dtype == torch.bool and torch.equal(keep, keep_ref)
    assert torch.equal(masked, masked_ref)
    assert int(keep[0].sum()) > 300 and int(keep[3].sum()) == 0 and not bool(keep[~own].any())
    # the rejection did something, and so did the second erosion
    assert int(keep.sum()) < int(((acc > 0.85) & ~obs.isnan() & own).sum()) - 200
