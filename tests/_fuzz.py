"""Randomised parity sweep on the GPU, shared by ``tests/test_gpu_parity.py`` (a fixed-seed slice in the driver-run
suite) and ``scripts/fuzz_parity.py`` (long runs): random scene sizes, image sizes, tile sizes, channel counts, view
counts and kernel-variant settings; every case renders forward + backward through ``rasterization()`` and is checked
against the CPU oracle (images 1e-4 off ambiguous pixels, gradients 2e-3 relative)."""
import os
import random
import re
import time

import torch

from oracle import gsplat_ref as R   # test infrastructure: the oracle is the checker here
from pogs_amd import _lib, rendering
from tests._util import assert_grad_close, assert_image_close, scene_args, small_scene


class FuzzFailure(AssertionError):
    pass


def run_cases(cases, seed, dev, large=False, only=-1, log=print, oracle_chunk=32):
    """Runs ``cases`` random cases; raises FuzzFailure on the first real mismatch.  Returns
    {"ok", "near", "ties", "seconds"}: ``near`` = cases with one gradient element within 3x of the bound (float32
    accumulation against a float64 oracle), ``ties`` = cases skipped because two depths closer than float32
    resolves sort differently in the float64 oracle.  The kernel-variant overrides are left at their last values:
    callers restore them (``tests/conftest.py::tuning`` does)."""
    rng = random.Random(seed)
    t0 = time.time()
    near = ties = ok = 0
    for case in range(cases):
        n = rng.choice([300, 900, 3000] if large else [1, 7, 64, 300, 900])
        w, h = rng.choice([(200, 120), (320, 200), (400, 300)] if large
                          else [(16, 16), (33, 17), (70, 45), (128, 96), (200, 120)])
        views = rng.choice([1, 1, 2, 3])
        ts = rng.choice([8, 16])
        F = rng.choice([0, 0, 5, 16, 29, 32, 61, 64, 93])
        mode = rng.choice(["RGB", "RGB+ED", "RGB+D"]) if F < 90 else "RGB"
        if n == 1 and mode == "RGB+ED":
            mode = "RGB+D"   # one Gaussian: depth/alpha is constant, its gradient is pure float32 cancellation residue
        freeze = rng.random() < 0.3
        env = {"POGS_RAST_FWD_PPL": rng.choice(["0", "1", "2", "4"]),
               "POGS_RAST_BWD_PPL": rng.choice(["0", "1", "2", "4"]),
               "POGS_RAST_FWD_MFMA": rng.choice(["0", "1"]), "POGS_RAST_BWD_MFMA": rng.choice(["0", "1", "3", "4", "5", "6", None]),
               "POGS_RAST_FWD_BLOCKS": rng.choice(["0", "0", "1", "2"]), "POGS_RAST_FWD_CHUNK": rng.choice([None, None, "100", "16"])}
        seed_scene = rng.randrange(10**6)
        if only >= 0 and case != only:
            continue
        _lib.set_tuning(**env)
        desc = f"case {case}: n={n} {w}x{h} views={views} ts={ts} F={F} {mode} freeze={freeze} {env}"
        sc = small_scene(n=n, w=w, h=h, views=views, seed=seed_scene, f=max(F, 1), lo=0.02, hi=0.35)
        a = scene_args(sc, device=dev)
        b = scene_args(sc, dtype=torch.float64)
        leaves_g = {k: a[k].clone().requires_grad_(not (freeze and k in ("colors", "features")))
                    for k in ("means", "quats", "scales", "opacities", "colors", "features")}
        leaves_c = {k: b[k].clone().requires_grad_(True) for k in leaves_g}
        kw = dict(sh_degree=3, render_mode=mode, packed=False, tile_size=ts, absgrad=True)
        out, alpha, meta = rendering.rasterization(leaves_g["means"], leaves_g["quats"], leaves_g["scales"],
                                                   leaves_g["opacities"], leaves_g["colors"], a["viewmats"], a["Ks"],
                                                   w, h, features=leaves_g["features"] if F else None, **kw)
        ref, ralpha, rmeta = R.rasterization(leaves_c["means"], leaves_c["quats"], leaves_c["scales"],
                                             leaves_c["opacities"], leaves_c["colors"], b["viewmats"], b["Ks"], w, h,
                                             features=leaves_c["features"] if F else None,
                                             radii_override=meta["radii"].cpu(), sh_degree=3, render_mode=mode,
                                             channel_chunk=oracle_chunk, tile_size=ts)
        amb = rmeta["ambiguous"]
        if not torch.equal(meta["flatten_ids"].cpu().long(), rmeta["flatten_ids"].long()):
            # two Gaussians whose depths differ by less than float32 resolves order differently in the float64
            # oracle: same alpha, same depth, colours off by alpha_a * alpha_b * T * (c_a - c_b) -- not comparable
            log(f"depth near-tie (order differs from the float64 oracle), skipped: {desc}")
            ties += 1
            continue
        if only >= 0:
            err = (out.detach().double().cpu() - ref.detach()).abs()
            err[amb] = 0
            i = int(err.argmax())
            c, y, x, ch = [int(v) for v in torch.unravel_index(torch.tensor(i), err.shape)]
            log(f"{desc}\nworst unflagged: {(c, y, x, ch)} err {float(err.max())} got {float(out[c, y, x, ch])} "
                f"ref {float(ref[c, y, x, ch])} alpha got/ref {float(alpha[c, y, x, 0])} {float(ralpha[c, y, x, 0])}"
                f"\nper-channel max err {[round(float(v), 5) for v in err.amax(dim=(0, 1, 2))]}"
                f"\nbad pixels per camera {[(int((err[ci].amax(-1) > 1e-4).sum())) for ci in range(err.shape[0])]}"
                f" n_isects {meta.get('n_composited_isects')}")
        try:
            out_cmp = out.detach().double().cpu()
            if mode == "RGB+ED":
                # depth / alpha turns an alpha flip at the 1/255 threshold into a jump of the whole depth value:
                # flagged pixels of the depth channel are not comparable (the other channels and all unflagged
                # pixels are)
                out_cmp[..., -1][amb] = ref.detach()[..., -1][amb]
            # the split-bf16 forward variant is held to its documented bound, 1.5e-5 * max|colour| (>= 1e-4)
            tol = 1e-4
            if env["POGS_RAST_FWD_MFMA"] == "1":
                tol = max(1e-4, 1.5e-5 * float(ref.detach().abs().max()))
            assert_image_close(out_cmp, ref, amb, "image", tol=tol)
            assert_image_close(alpha, ralpha, amb, "alpha")
            g = torch.Generator().manual_seed(case)
            keep = (~amb)[..., None].float()
            vc, va = torch.randn(ref.shape, generator=g) * keep, torch.randn(ralpha.shape, generator=g) * keep
            ((out * vc.to(dev)).sum() + (alpha * va.to(dev)).sum()).backward()
            if ref.requires_grad or ralpha.requires_grad:
                ((ref * vc.double()).sum() + (ralpha * va.double()).sum()).backward()
            else:  # everything culled: the oracle returns constants; the device path must hand back zero gradients
                assert all(v.grad is None or float(v.grad.abs().max()) == 0.0 for v in leaves_g.values()), "culled grads"
            ca = rmeta.get("colour_clamp_ambiguous")
            if ca is not None and bool(ca.any()) and leaves_g["colors"].grad is not None:
                # a view-dependent colour within float32 resolution of the clamp at zero: that (Gaussian, channel)'s
                # coefficient gradient is either everything or nothing; so is what reaches the mean through the
                # view direction
                keep_c = (~ca)[:, None, :].double()
                leaves_c["colors"].grad.mul_(keep_c)
                leaves_g["colors"].grad.mul_(keep_c.float().to(dev))
                rows = ca.any(dim=-1)
                for k in ("means",):
                    if leaves_c[k].grad is not None and leaves_g[k].grad is not None:
                        leaves_c[k].grad[rows] = 0
                        leaves_g[k].grad[rows.to(dev)] = 0
            for k in leaves_g:
                if leaves_c[k].grad is not None and leaves_g[k].requires_grad and (k != "features" or F):
                    # twice the unit-test tolerance: float32 accumulation over up to 97 channels x whole images
                    assert_grad_close(leaves_g[k].grad, leaves_c[k].grad, k, rtol=2e-3, atol_frac=2e-4)
        except AssertionError as e:
            # a single element a little over the bound is float32 round-off against the float64 oracle (hundreds of
            # Gaussians per pixel in the densest cases); a wrong kernel is off by orders of magnitude
            m = re.search(r"worst ratio ([0-9.eE+-]+|nan|inf)", str(e))
            ratio = float(m.group(1)) if m else float("inf")
            if ratio == ratio and ratio < 3.0:
                log(f"near-miss {desc}\n    {str(e)[:200]}")
                near += 1
                continue
            raise FuzzFailure(f"{desc}\n    {str(e)[:300]}") from e
        ok += 1
        if (case % 10 == 9 and cases > 200) or (large and case % 5 == 4):
            log(f"{case + 1} cases ok, {time.time() - t0:.0f} s")
    return {"ok": ok, "near": near, "ties": ties, "seconds": time.time() - t0}
