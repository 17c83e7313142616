"""Helpers shared by the parity tests: small seeded scenes and the comparison rules.

Comparison rules (the bar of the north star):
  * tile counts, intersection ids, sort order, offsets: bit-exact;
  * colour / alpha / features: max-abs error <= 1e-4 on every pixel that is not within a hair of one of
    the compositing discontinuities (the oracle flags those, see ``oracle.gsplat_ref._composite``);
    flagged pixels must stay a tiny fraction and within the size of one skipped contribution;
  * gradients: elementwise |d| <= 1e-3 |ref| + 1e-4 max|ref|  (1e-3 relative).
"""
import torch

from pogs_amd.scenes import make_scene

IMG_TOL = 1e-4
GRAD_RTOL = 1e-3


def small_scene(n=300, w=64, h=48, f=5, views=1, seed=1, lo=0.02, hi=0.25):
    return make_scene(n, w, h, n_features=f, n_views=views, seed=seed, scale_lo=lo, scale_hi=hi)


def scene_args(sc, dtype=torch.float32, device="cpu", requires_grad=False):
    a = dict(
        means=sc.means, quats=sc.quats, scales=sc.log_scales.exp(), opacities=torch.sigmoid(sc.opacity_logits),
        colors=sc.sh, viewmats=sc.viewmats, Ks=sc.Ks, features=sc.features)
    out = {}
    for k, v in a.items():
        t = v.to(dtype).to(device).contiguous()
        if requires_grad and k not in ("Ks",):
            t.requires_grad_(True)
        out[k] = t
    out["width"], out["height"] = sc.width, sc.height
    return out


# (name, worst |got - ref| / |ref| over the elements with |ref| > 1e-3 max|ref|, elements counted): what the
# comparisons below measured beyond what they assert; tests/conftest.py writes it out at the end of a session.
REL_ERRORS = []
# (name, largest error / allowed error) of every gradient and image comparison: how close each came to its bound
MARGINS = []


# (name, HIP kernel's worst relative error against the float64 oracle, the float32 C restatement's -- gsplat's operation
# order, scalar loops -- on the same inputs and elements, elements counted): what float32 itself costs, next to what
# the kernel adds
PRECISION_PAIRS = []


def worst_relative_error(got, ref, floor=1e-3):
    """max |got - ref| / |ref| over the elements with |ref| > floor * max |ref|  (-> (value, elements counted))."""
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    big = ref.abs() > floor * ref.abs().max()
    if not big.any():
        return 0.0, 0
    return float(((got - ref).abs()[big] / ref.abs()[big]).max()), int(big.sum())


def record_precision(name, hip, c32, ref64):
    h, n = worst_relative_error(hip, ref64)
    c, _ = worst_relative_error(c32, ref64)
    PRECISION_PAIRS.append((name, h, c, n))
    return h, c


def assert_grad_close(got, ref, name, rtol=GRAD_RTOL, atol_frac=1e-4):
    if ref is None:  # the oracle's graph does not reach this input: the gradient is identically zero
        assert got is None or got.abs().max().item() == 0, name
        return
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = ref.abs().max().item()
    if scale == 0:
        assert got.abs().max().item() == 0, name
        return
    err = (got - ref).abs()
    big = ref.abs() > 1e-3 * scale
    if big.any():
        REL_ERRORS.append((name, float((err[big] / ref.abs()[big]).max()), int(big.sum())))
    bound = rtol * ref.abs() + atol_frac * scale
    MARGINS.append((name, float((err / bound).max())))
    bad = err > bound
    assert not bad.any(), (
        f"{name}: {int(bad.sum())}/{bad.numel()} elements off; max err {err.max().item():.3e} "
        f"(scale {scale:.3e}, worst ratio {(err / bound).max().item():.2f})")


def assert_image_close(got, ref, ambiguous, name, tol=IMG_TOL):
    """got/ref [C,H,W,D]; ambiguous [C,H,W] bool."""
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    err = (got - ref).abs().amax(dim=-1)
    clean = ~ambiguous
    worst = err[clean].max().item() if clean.any() else 0.0
    MARGINS.append((name, worst / tol))
    assert worst <= tol, f"{name}: max abs err {worst:.3e} > {tol} on unflagged pixels"
    frac = ambiguous.float().mean().item()
    # at most 2 % of the pixels may be flagged -- or 8 pixels: in a 16x16 image every pixel is 0.4 %
    assert frac <= max(0.02, 8.0 / ambiguous.numel()), f"{name}: {frac:.3%} of pixels flagged ambiguous"
    if ambiguous.any():
        scale = max(1.0, ref.abs().max().item())
        assert err[ambiguous].max().item() <= 2e-2 * scale, f"{name}: flagged pixel off by more than one splat"


def relu_ambiguous(x, al, ws, rel=2e-4):
    """Pixels where some hidden pre-activation lies within the forward error bound of zero.  The gradient through a
    ReLU is a step function of its pre-activation, so at such a pixel a float32 / split-bf16 forward pass may take the
    other branch than the float64 one and the whole pixel's contribution differs by O(1): the gradient tests give
    those pixels a zero cotangent (for both sides) and compare the rest to 1e-3."""
    h = x.double()
    if al is not None:
        a = al.double()
        h = torch.where(a > 0, h / a.clamp(min=1e-6), torch.zeros_like(h))
    amb = torch.zeros(h.shape[0], dtype=torch.bool)
    for w in ws[:-1]:
        pre = h @ w.double().t()
        amb |= (pre.abs() < rel * pre.abs().amax(-1, keepdim=True).clamp(min=1.0)).any(-1)   # error ~ the pixel's own magnitude
        h = torch.relu(pre)
    return amb
