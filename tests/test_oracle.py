"""CPU tests of the oracle itself.  The reference pins nothing at the rasterizer boundary ("parity
unpinned"), so the oracle is pinned by (1) closed-form known answers, (2) agreement between three
independently written restatements (float64 tile-free brute force, tile lists, C scalar loops),
(3) finite differences, (4) committed golden vectors (regression pin for every later round)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import gsplat_ref as R
from oracle import raster_c as RC
from tests._util import scene_args, small_scene

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _one(xy, conic, opac, color, W=32, H=32, ts=16, dt=torch.float64):
    """Rasterize explicit 2-D Gaussians (already projected) through the tile-list compositor."""
    n = len(xy)
    means2d = torch.tensor(xy, dtype=dt)[None]
    conics = torch.tensor(conic, dtype=dt)[None]
    opacities = torch.tensor(opac, dtype=dt)[None]
    colors = torch.tensor(color, dtype=dt)[None]
    radii = torch.full((1, n), 40, dtype=torch.int32)
    depths = torch.arange(1, n + 1, dtype=dt)[None]
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    _, ids, flat = R.isect_tiles(means2d, radii, depths, ts, tw, th)
    offs = R.isect_offset_encode(ids, 1, tw, th)
    return R.rasterize_to_pixels(means2d, conics, colors, opacities, W, H, ts, offs, flat)


def test_single_gaussian_closed_form():
    s2 = 9.0
    out, alpha = _one([[16.0, 16.0]], [[1 / s2, 0.0, 1 / s2]], [0.8], [[0.25, 0.5, 1.0]])
    yy, xx = torch.meshgrid(torch.arange(32.0, dtype=torch.float64), torch.arange(32.0, dtype=torch.float64),
                            indexing="ij")
    r2 = (xx + 0.5 - 16) ** 2 + (yy + 0.5 - 16) ** 2
    a = 0.8 * torch.exp(-0.5 * r2 / s2)
    a = torch.where(a >= 1 / 255, a, torch.zeros_like(a))
    assert torch.allclose(alpha[0, ..., 0], a, atol=1e-14)
    assert torch.allclose(out[0, ..., 2], a, atol=1e-14) and torch.allclose(out[0, ..., 0], 0.25 * a, atol=1e-14)


def test_two_gaussians_order_and_transmittance():
    # the nearer (first in depth) Gaussian occludes the second: C = c1 a1 + c2 a2 (1 - a1)
    out, alpha = _one([[16.5, 16.5], [16.5, 16.5]], [[0.1, 0, 0.1]] * 2, [0.6, 0.9], [[1.0, 0.0], [0.0, 1.0]])
    assert out[0, 16, 16, 0].item() == pytest.approx(0.6)
    assert out[0, 16, 16, 1].item() == pytest.approx(0.9 * 0.4)
    assert alpha[0, 16, 16, 0].item() == pytest.approx(1 - 0.4 * 0.1)


def test_alpha_clamp_and_stop_rule():
    # opacity 1 clamps to 0.999; the second such Gaussian would leave T = 1e-6 <= 1e-4 -> it is NOT blended
    out, alpha = _one([[16.5, 16.5]] * 3, [[0.1, 0, 0.1]] * 3, [1.0, 1.0, 1.0], [[1.0], [10.0], [100.0]])
    assert alpha[0, 16, 16, 0].item() == pytest.approx(0.999)
    assert out[0, 16, 16, 0].item() == pytest.approx(0.999)
    # with 0.995: T = 5e-3, then 2.5e-5 <= 1e-4 stops before the second one as well
    out, alpha = _one([[16.5, 16.5]] * 2, [[0.1, 0, 0.1]] * 2, [0.995, 0.995], [[1.0], [10.0]])
    assert alpha[0, 16, 16, 0].item() == pytest.approx(0.995)
    # with 0.98: T = 0.02 then 4e-4 > 1e-4 -> both blended
    out, alpha = _one([[16.5, 16.5]] * 2, [[0.1, 0, 0.1]] * 2, [0.98, 0.98], [[1.0], [10.0]])
    assert alpha[0, 16, 16, 0].item() == pytest.approx(1 - 4e-4)


def test_small_alpha_is_skipped_and_negative_sigma_rejected():
    out, alpha = _one([[16.5, 16.5]], [[0.1, 0, 0.1]], [1 / 256], [[1.0]])
    assert alpha.abs().max().item() == 0
    # indefinite conic: sigma < 0 off-axis -> those pixels are skipped
    out, alpha = _one([[16.5, 16.5]], [[0.1, 0.5, 0.1]], [0.9], [[1.0]])
    assert alpha[0, 10, 22, 0].item() == 0 and alpha[0, 16, 16, 0].item() == pytest.approx(0.9)
    # same pixel without the cross term is well inside the footprint
    out, alpha = _one([[16.5, 16.5]], [[0.01, 0.0, 0.01]], [0.9], [[1.0]])
    assert alpha[0, 10, 22, 0].item() > 0.4


def test_projection_culling_and_closed_form():
    dt = torch.float64
    means = torch.tensor([[0, 0, 0.0], [0, 0, -3.5], [50.0, 0, 0], [0, 0, 0.0]], dtype=dt)
    quats = torch.tensor([[1.0, 0, 0, 0]] * 4, dtype=dt)
    scales = torch.full((4, 3), 0.1, dtype=dt)
    viewmat = torch.eye(4, dtype=dt)[None].clone()
    viewmat[0, 2, 3] = 3.0
    K = torch.tensor([[[100.0, 0, 32], [0, 100.0, 24], [0, 0, 1]]], dtype=dt)
    radii, m2, dep, con, comp = R.fully_fused_projection(means, quats, scales, viewmat, K, 64, 48,
                                                         calc_compensations=True)
    assert radii[0, 1] == 0 and radii[0, 2] == 0          # behind the near plane; off-screen
    assert radii[0, 0] > 0 and torch.allclose(m2[0, 0], torch.tensor([32.0, 24.0], dtype=dt))
    var = (0.1 * 100 / 3) ** 2                            # isotropic: sigma_px = s f / z
    assert con[0, 0, 0].item() == pytest.approx(1 / (var + 0.3)) and abs(con[0, 0, 1].item()) < 1e-12
    assert radii[0, 0].item() == math.ceil(3 * math.sqrt(var + 0.3 + 0.1))   # sqrt(max(b^2-det, .01)) = .1
    assert comp[0, 0].item() == pytest.approx(var / (var + 0.3))
    assert dep[0, 0].item() == pytest.approx(3.0)


def test_sh_basis_is_orthonormal():
    g = torch.Generator().manual_seed(0)
    d = torch.randn(400_000, 3, generator=g, dtype=torch.float64)
    d = d / d.norm(dim=-1, keepdim=True)
    B = R.eval_sh_bases(16, d)
    gram = 4 * math.pi * (B.T @ B) / d.shape[0]
    assert torch.allclose(gram, torch.eye(16, dtype=torch.float64), atol=1.5e-2)
    assert B[:, 0].min().item() == pytest.approx(0.2820947917738781)


def test_isect_small_example_by_hand():
    # one Gaussian centred at (20,20) radius 6, tile 16 on a 3x2 grid: rect x [0,2), y [0,2) -> 4 tiles
    means2d = torch.tensor([[[20.0, 20.0], [40.0, 8.0]]])
    radii = torch.tensor([[6, 3]], dtype=torch.int32)
    depths = torch.tensor([[2.0, 1.0]])
    tpg, ids, flat = R.isect_tiles(means2d, radii, depths, 16, 3, 2)
    assert tpg.tolist() == [[4, 1]]
    tb = (3 * 2).bit_length()
    tiles = ((ids >> 32) & ((1 << tb) - 1)).tolist()
    assert tiles == [0, 1, 2, 3, 4] and flat.tolist() == [0, 0, 1, 0, 0]
    assert (ids & 0xFFFFFFFF).tolist()[2] == np.float32(1.0).view(np.int32)
    offs = R.isect_offset_encode(ids, 1, 3, 2)
    assert offs.flatten().tolist() == [0, 1, 2, 3, 4, 5]


def test_tile_lists_agree_with_tile_free_brute_force():
    sc = small_scene(n=300, w=64, h=48, views=2, seed=1)
    a = scene_args(sc, dtype=torch.float64)
    kw = dict(sh_degree=3, render_mode="RGB+ED")
    for ts in (8, 16):
        c1, a1, _ = R.rasterization(**a, **kw, tile_size=ts)
        c2, a2, _ = R.rasterization(**a, **kw, tile_size=ts, bruteforce=True)
        assert (c1 - c2).abs().max().item() < 1e-12 and (a1 - a2).abs().max().item() < 1e-12


def _staged(sc, ts, D, dt=torch.float64):
    a = scene_args(sc, dtype=dt)
    C = sc.viewmats.shape[0]
    radii, m2, dep, con, _ = R.fully_fused_projection(a["means"], a["quats"], a["scales"], a["viewmats"], a["Ks"],
                                                      sc.width, sc.height)
    tw, th = math.ceil(sc.width / ts), math.ceil(sc.height / ts)
    _, ids, flat = R.isect_tiles(m2, radii, dep, ts, tw, th)
    offs = R.isect_offset_encode(ids, C, tw, th)
    g = torch.Generator().manual_seed(0)
    cols = torch.randn(C, sc.n, D, generator=g, dtype=dt)
    op = a["opacities"][None].expand(C, -1).contiguous()
    bg = torch.rand(C, D, generator=g, dtype=dt)
    return m2, con, cols, op, bg, offs, flat


@pytest.mark.parametrize("ts", [8, 16])
def test_c_restatement_agrees_with_torch_oracle(ts):
    sc = small_scene(n=600, w=70, h=45, views=2, seed=4, lo=0.02, hi=0.3)
    m2, con, cols, op, bg, offs, flat = _staged(sc, ts, 7)
    lv = [t.clone().requires_grad_(True) for t in (m2, con, cols, op)]
    o, a, amb = R.rasterize_to_pixels(*lv, 70, 45, ts, offs, flat, backgrounds=bg, return_ambiguous=True)
    oc, ac, last = RC.raster_fwd(m2, con, cols, op, 70, 45, ts, offs, flat, backgrounds=bg)
    keep = (~amb)[..., None]
    assert ((oc.double() - o).abs() * keep).max().item() < 1e-4
    assert ((ac.double() - a).abs() * keep).max().item() < 1e-5
    g = torch.Generator().manual_seed(5)
    vc = torch.randn(o.shape, generator=g, dtype=torch.float64) * keep
    va = torch.randn(a.shape, generator=g, dtype=torch.float64) * keep
    ((o * vc).sum() + (a * va).sum()).backward()
    gm, gma, gcn, gcl, gop = RC.raster_bwd(m2, con, cols, op, 70, 45, ts, offs, flat, ac, last, vc, va, backgrounds=bg)
    for got, ref in ((gm, lv[0].grad), (gcn, lv[1].grad), (gcl, lv[2].grad), (gop, lv[3].grad)):
        assert (got.double() - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    assert (gma + 1e-6 >= gm.abs()).all()


def test_oracle_gradients_match_finite_differences():
    sc = small_scene(n=40, w=32, h=32, views=1, seed=9, lo=0.05, hi=0.3)
    a = scene_args(sc, dtype=torch.float64)
    g = torch.Generator().manual_seed(2)
    v = None

    def f(means, scales, opac):
        nonlocal v
        c, al, _ = R.rasterization(**{**a, "means": means, "scales": scales, "opacities": opac}, sh_degree=3,
                                   render_mode="RGB+ED")
        if v is None:
            v = torch.randn(c.shape, generator=g, dtype=torch.float64)
        return (c * v).sum() + al.sum()

    x = [a["means"].clone().requires_grad_(True), a["scales"].clone().requires_grad_(True),
         a["opacities"].clone().requires_grad_(True)]
    f(*x).backward()
    eps = 1e-6
    for ti, t in enumerate(x):
        flat = t.detach().flatten()
        for j in (0, 7, flat.numel() - 1):
            d = torch.zeros_like(flat)
            d[j] = eps
            args_p = [u.detach() + (d.view_as(u) if k == ti else 0) for k, u in enumerate(x)]
            args_m = [u.detach() - (d.view_as(u) if k == ti else 0) for k, u in enumerate(x)]
            fd = (f(*args_p) - f(*args_m)).item() / (2 * eps)
            an = t.grad.flatten()[j].item()
            assert fd == pytest.approx(an, rel=2e-4, abs=1e-6), (ti, j)


# ------------------------------------------------------------------------------------------------
# golden vectors (generated by tests/golden/make_golden.py from the float64 oracle)
# ------------------------------------------------------------------------------------------------
def _golden_cases():
    return sorted(f for f in os.listdir(GOLDEN) if f.endswith(".npz")) if os.path.isdir(GOLDEN) else []


def test_golden_files_present():
    assert len(_golden_cases()) >= 4


@pytest.mark.parametrize("case", _golden_cases())
def test_oracle_reproduces_golden(case):
    from tests.golden.make_golden import run_case

    z = np.load(os.path.join(GOLDEN, case), allow_pickle=False)
    spec = {k[5:]: z[k].item() for k in z.files if k.startswith("spec_")}
    res = run_case(spec, torch.float64)
    for k in ("render_colors", "render_alphas", "v_means", "v_quats", "v_scales", "v_opacities", "v_colors"):
        ref = torch.from_numpy(z[k])
        assert torch.allclose(res[k], ref, rtol=1e-9, atol=1e-11), (case, k)
    assert np.array_equal(res["isect_ids"].numpy(), z["isect_ids"])
    assert np.array_equal(res["flatten_ids"].numpy(), z["flatten_ids"])
    assert np.array_equal(res["isect_offsets"].numpy(), z["isect_offsets"])
    # the float32 run of the same oracle stays within the image tolerance of the golden
    r32 = run_case(spec, torch.float32)
    keep = ~torch.from_numpy(z["ambiguous"])[..., None]
    if np.array_equal(r32["isect_ids"].numpy() >> 32, z["isect_ids"] >> 32):
        assert ((r32["render_colors"].double() - torch.from_numpy(z["render_colors"])).abs() * keep).max() <= 1e-4
