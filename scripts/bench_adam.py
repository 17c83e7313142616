"""Optimiser step at the benchmark model size (1M Gaussians: means, SH dc/rest, opacities, scales, quats and
32 feature channels = 91M parameters): pr_adam_step (one launch) against torch.optim.Adam per group.
Algorithmic bytes: 28 per parameter (read p, g, m, v; write p, m, v)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd.optim import Optimizers  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
shapes = {"means": (N, 3), "features_dc": (N, 3), "features_rest": (N, 15, 3), "opacities": (N, 1), "scales": (N, 3),
          "quats": (N, 4), "dino_feats": (N, 32)}
lrs = {"means": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacities": 0.05, "scales": 0.005,
       "quats": 0.001, "dino_feats": 1e-2}
params = {k: [torch.randn(s, device=dev).requires_grad_(True)] for k, s in shapes.items()}
for ps in params.values():
    ps[0].grad = torch.randn_like(ps[0])
n_par = sum(p[0].numel() for p in params.values())


def timed(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


fused = Optimizers({k: {"lr": v, "eps": 1e-15} for k, v in lrs.items()}, params)
ms_fused = timed(fused.optimizer_step_all)
ref = [torch.optim.Adam(params[k], lr=lrs[k], eps=1e-15) for k in shapes]
ms_torch = timed(lambda: [o.step() for o in ref])
print(json.dumps({"parameters": n_par, "pr_adam_step_ms": ms_fused, "GBps": n_par * 28 / ms_fused / 1e6,
                  "frac_of_8TBps": n_par * 28 / ms_fused / 1e6 / 8000, "torch_adam_per_group_ms": ms_torch}))
