"""The host glue either side of the rasterizer call, driven the way ``POGSModel.get_outputs`` drives it
(``pogs/pogs.py:247-524``) on the reference's own camera convention: nerfstudio camera-to-world poses and intrinsics of
the bundled ``box`` dataset (``tests/golden/box_cameras.json``, extracted by ``tests/golden/make_box_cameras.py``)
-> ``get_viewmat`` (``pogs/pogs.py:75-90``) -> ``rasterization`` RGB+ED (``:352-371``) -> background composite and clamp
(``:382-383``) -> depth fill (``:386-387``) -> feature render (``:493-510``) -> alpha normalisation + channel MLP
(``:516-522``).  Checked against the float64 oracle doing the same operations with torch CPU operators."""
import json
import os

import pytest
import torch

from oracle import gsplat_ref as R
from oracle.mlp_ref import feature_mlp_ref
from tests._util import assert_grad_close, assert_image_close, relu_ambiguous

HERE = os.path.dirname(os.path.abspath(__file__))


def _box_cameras(scale_down=4):
    z = json.load(open(os.path.join(HERE, "golden", "box_cameras.json")))
    c2w = torch.tensor([f["transform_matrix"] for f in z["frames"]], dtype=torch.float64)[:, :3, :]
    Ks = torch.zeros(len(z["frames"]), 3, 3, dtype=torch.float64)
    for i, f in enumerate(z["frames"]):   # nerfstudio rescales the intrinsics with the image (cameras.py rescale_output_resolution)
        Ks[i] = torch.tensor([[f["fl_x"] / scale_down, 0, f["cx"] / scale_down],
                              [0, f["fl_y"] / scale_down, f["cy"] / scale_down], [0, 0, 1.0]])
    return c2w, Ks, z["frames"][0]["w"] // scale_down, z["frames"][0]["h"] // scale_down


def _viewmat_by_general_inverse(c2w):
    """Independent of get_viewmat's analytic inverse: flip the camera's y and z axes (OpenGL -> OpenCV), invert."""
    m = torch.eye(4, dtype=c2w.dtype).repeat(c2w.shape[0], 1, 1)
    m[:, :3, :] = c2w
    m[:, :3, 1:3] *= -1
    return torch.linalg.inv(m)


def _look_at_point(c2w):
    """Least-squares meeting point of the cameras' optical axes (OpenGL: the camera looks along -z)."""
    o, d = c2w[:, :, 3], -c2w[:, :, 2]
    A = torch.eye(3, dtype=c2w.dtype)[None] - d[:, :, None] * d[:, None, :]
    return torch.linalg.solve(A.sum(0), (A @ o[:, :, None]).sum(0)).flatten()


def test_box_cameras_fixture_is_a_set_of_rigid_poses():
    c2w, Ks, W, H = _box_cameras()
    assert c2w.shape == (4, 3, 4) and (W, H) == (160, 120)
    Rm = c2w[:, :, :3]
    assert torch.allclose(Rm @ Rm.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(4, 3, 3), atol=1e-5)
    from pogs_amd.scenes import get_viewmat

    assert torch.allclose(get_viewmat(c2w), _viewmat_by_general_inverse(c2w), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
def test_get_outputs_glue_on_box_cameras(gpu, training):
    from pogs_amd import rendering as rd
    from pogs_amd.mlp import FeatureMLP
    from pogs_amd.scenes import composite_background, fill_depth, get_viewmat, normalise_features, rgb2sh

    c2w, Ks, W, H = _box_cameras()
    centre = _look_at_point(c2w)
    g = torch.Generator().manual_seed(4)
    n = 500
    P64 = dict(
        means=centre[None] + torch.randn(n, 3, generator=g, dtype=torch.float64) * 0.06,
        quats=torch.randn(n, 4, generator=g, dtype=torch.float64),
        log_scales=torch.log(torch.rand(n, 3, generator=g, dtype=torch.float64) * 0.02 + 0.004),
        logits=torch.randn(n, generator=g, dtype=torch.float64) * 1.5,
        sh=torch.cat([rgb2sh(torch.rand(n, 1, 3, generator=g, dtype=torch.float64)),
                      torch.randn(n, 15, 3, generator=g, dtype=torch.float64) * 0.1], dim=1),
        dino=torch.randn(n, 64, generator=g, dtype=torch.float64))
    net = FeatureMLP(device=gpu)
    net.train(training)
    background = torch.tensor([0.2, 0.5, 0.7], dtype=torch.float64)
    g2 = torch.Generator().manual_seed(9)
    cot = {"rgb": torch.randn(H, W, 3, generator=g2, dtype=torch.float64),
           "depth": torch.randn(H, W, 1, generator=g2, dtype=torch.float64),
           "dino": torch.randn(H, W, 64, generator=g2, dtype=torch.float64)}

    def get_outputs(p, cam, rasterization, dev, dtype, mlp, extra):
        """pogs/pogs.py:247-524 for one camera (crop box, obj_id gather and viewer branches left out)."""
        viewmat = get_viewmat(c2w[cam:cam + 1].to(dtype).to(dev))
        K = Ks[cam:cam + 1].to(dtype).to(dev)
        scales, opac = torch.exp(p["log_scales"]), torch.sigmoid(p["logits"])          # :355-356
        render, alpha, info = rasterization(
            means=p["means"], quats=p["quats"], scales=scales, opacities=opac, colors=p["sh"], viewmats=viewmat, Ks=K,
            width=W, height=H, tile_size=16, packed=False, near_plane=0.01, far_plane=1e10, render_mode="RGB+ED",
            sh_degree=3, sparse_grad=False, absgrad=True, rasterize_mode="classic", **extra(info_of="rgb"))
        rgb = composite_background(render, alpha, background.to(dtype).to(dev))        # :382-383
        depth = fill_depth(render, alpha)                                              # :386-387
        feats, falpha, finfo = rasterization(                                          # :493-510: geometry detached
            means=p["means"].detach(), quats=p["quats"].detach(), scales=scales.detach(), opacities=opac.detach(),
            colors=p["dino"], viewmats=viewmat, Ks=K, width=W, height=H, tile_size=16, packed=False, near_plane=0.01,
            far_plane=1e10, render_mode="RGB", sparse_grad=False, absgrad=False, rasterize_mode="classic",
            **extra(info_of="dino"))
        return {"rgb": rgb.squeeze(0), "depth": depth.squeeze(0), "dino": mlp(feats, falpha).squeeze(0), "feats": feats,
                "alpha": alpha.squeeze(0), "dino_alpha": falpha.squeeze(0), "info": info, "finfo": finfo}

    for cam in (0, 2):
        pg = {k: v.float().to(gpu).requires_grad_(True) for k, v in P64.items()}
        pc = {k: v.clone().requires_grad_(True) for k, v in P64.items()}
        out = get_outputs(pg, cam, rd.rasterization, gpu, torch.float32, net, lambda info_of: {})
        radii = {"rgb": out["info"]["radii"].cpu(), "dino": out["finfo"]["radii"].cpu()}
        ws = [w.detach().double().cpu().requires_grad_(training) for w in net.linear_weights()]
        wref = {f"mlp weight {i}": w for i, w in enumerate(ws)}
        ref = get_outputs(pc, cam, R.rasterization, "cpu", torch.float64,
                          lambda f, a: feature_mlp_ref(f, a, ws, 0.0 if training else FeatureMLP.EVAL_ALPHA_CUT),
                          lambda info_of: {"radii_override": radii[info_of]})
        assert int((out["info"]["radii"] > 0).sum()) > 300 and 0.2 < float(ref["alpha"].detach().mean()) < 0.95
        amb = ref["info"]["ambiguous"][0] | ref["finfo"]["ambiguous"][0]
        # the 0.8 alpha cut and the alpha > 0 tests are steps: pixels within 1e-4 of them may fall either side
        if not training:
            amb = amb | ((ref["dino_alpha"][..., 0] - FeatureMLP.EVAL_ALPHA_CUT).abs() < 1e-4)
        assert_image_close(out["rgb"][None], ref["rgb"][None], amb[None], f"rgb cam {cam}")
        empty = ref["alpha"][..., 0] == 0
        assert torch.equal(out["alpha"][..., 0].cpu() == 0, empty)
        d_got, d_ref = out["depth"].double().cpu(), ref["depth"]
        # empty pixels hold the image maximum of the depth channel (one value, taken over ambiguous pixels too)
        assert torch.allclose(d_got[empty], d_got[~empty].max().expand_as(d_got[empty]), rtol=0, atol=0) or not empty.any()
        keep = ~amb & ~empty
        assert (d_got[keep] - d_ref[keep]).abs().max().item() <= 1e-4
        # features after the MLP: 1e-4 of the output scale (split-bf16 layers) plus what a 1e-4 input error becomes
        scale = max(1.0, ref["dino"].abs().max().item())
        derr = (out["dino"].double().cpu() - ref["dino"]).abs().amax(-1)
        assert derr[~amb].max().item() <= 2e-4 * scale, derr[~amb].max().item()
        # normalise_features is the torch form of what the fused MLP does first
        nf = normalise_features(ref["dino_alpha"].new_ones(H, W, 64) * ref["dino_alpha"], ref["dino_alpha"])
        assert torch.allclose(nf[~empty], torch.ones(64, dtype=torch.float64).expand_as(nf[~empty])) and nf[empty].abs().sum() == 0
        if not training:
            continue
        keepc = (~amb)[..., None].double()
        # through the ReLUs backwards the gradient is a step function of the pre-activations: pixels with one of them
        # within the forward error of zero get no feature cotangent (either side may take either branch there)
        amb_relu = relu_ambiguous(ref["feats"].detach().reshape(-1, 64), ref["dino_alpha"].detach().reshape(-1, 1), ws).view(H, W)
        keep = {"rgb": keepc, "depth": keepc, "dino": (~(amb | amb_relu))[..., None].double()}
        loss_g = sum((out[k] * (cot[k] * keep[k]).float().to(gpu)).sum() for k in cot)
        loss_c = sum((ref[k] * cot[k] * keep[k]).sum() for k in cot)
        loss_g.backward()
        loss_c.backward()
        for k in ("means", "quats", "log_scales", "logits", "sh"):
            assert_grad_close(pg[k].grad, pc[k].grad, f"{k} cam {cam}")
        assert_grad_close(pg["dino"].grad, pc["dino"].grad, f"dino cam {cam}")
        for wg_, (name, wr_) in zip(net.linear_weights(), wref.items()):
            assert_grad_close(wg_.grad, wr_.grad, f"{name} cam {cam}")
            wg_.grad = None
