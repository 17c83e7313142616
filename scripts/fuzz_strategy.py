"""Randomised sweep of the densification refinement on the GPU against the CPU restatement: random set sizes,
thresholds (including "everything duplicated / split / pruned" extremes), radius rules, revised opacity.
Usage: fuzz_strategy.py [cases] [seed]"""
import copy
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import strategy_ref as R  # noqa: E402  (this script is a checker)
from tests.test_strategy import _assert_same, _device_copy, _scene  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
for case in range(cases):
    n = rng.choice([1, 2, 7, 8, 9, 63, 2047, 2048, 2049, 5000, 33333])
    cfg = R.Config(grow_grad2d=rng.choice([0.0, 0.0008, 0.002, 1e9]), grow_scale3d=rng.choice([0.0, 0.015, 1e9]),
                   prune_opa=rng.choice([0.0, 0.05, 0.5, 1.1]), prune_scale3d=rng.choice([0.0, 0.028, 1e9]),
                   reset_every=3000, refine_scale2d_stop_iter=rng.choice([0, 4000]),
                   revised_opacity=rng.random() < 0.3, grow_scale2d=rng.choice([0.05, 0.15, 1e9]),
                   prune_scale2d=rng.choice([0.1, 0.18, 1e9]))
    step = rng.choice([600, 3600, 4400])
    desc = f"case {case}: n={n} step={step} {cfg}"
    params, adam, stats = _scene(n, seed=rng.randrange(10**6))
    strat, p, opts, state = _device_copy(params, adam, stats, cfg, dev)
    ref_p, ref_adam = copy.deepcopy(params), copy.deepcopy(adam)
    ref_state = {"grad2d": stats["grad2d"].clone(), "count": stats["count"].clone(), "scene_scale": 1.0}
    if cfg.refine_scale2d_stop_iter > 0:
        ref_state["radii"] = stats["radii"].clone()

    def device_noise(k):
        torch.manual_seed(case)
        return torch.randn(2, k, 3, device=dev).cpu()

    want = R.grow(cfg, ref_p, ref_adam, ref_state, step, device_noise) + (R.prune(cfg, ref_p, ref_adam, ref_state, step),)
    torch.manual_seed(case)
    got = strat._refine(p, opts, state, step)
    try:
        assert got == want, (got, want)
        for k in ref_p:
            revised = k == "opacities" and cfg.revised_opacity
            _assert_same(k, p[k], ref_p[k], exact=k not in ("means", "scales") and not revised, tol=2e-3 if revised else 1e-6)
            if k in ref_adam:
                st = opts[k].state[p[k]]
                _assert_same(k + ".exp_avg", st["exp_avg"], ref_adam[k]["exp_avg"], exact=True)
                _assert_same(k + ".exp_avg_sq", st["exp_avg_sq"], ref_adam[k]["exp_avg_sq"], exact=True)
        for k in ("grad2d", "count"):
            _assert_same("state." + k, state[k], ref_state[k], exact=True)
    except AssertionError as e:
        print("FAIL", desc, "\n   ", str(e)[:300], flush=True)
        sys.exit(1)
    if case % 20 == 19:
        print(f"{case + 1} cases ok (last: {got}, {len(p['means'])} left)", flush=True)
print(f"all {cases} cases ok")
