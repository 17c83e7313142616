"""Per-entry-point GPU time of the tracker's per-object-crop branch (all crops in one call per pass), ms per iteration."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering  # noqa: E402
from pogs_amd.tracking import TrackingLoop  # noqa: E402

dev = torch.device("cuda:0")
loop = TrackingLoop.synthetic(dev, n=200_000, width=640, height=360, groups=4, seed=0, rois=True)
loop.frame(50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    loop.frame(50)
torch.cuda.synchronize()
print("ms/frame", (time.perf_counter() - t0) / 3 * 1e3, flush=True)
timer = rendering.StageTimer()
rendering.set_stage_timer(timer)
loop.frame(50)
torch.cuda.synchronize()
rendering.set_stage_timer(None)
st = {k: round(v[1] / 50, 4) for k, v in timer.summary().items()}
print(dict(sorted(st.items(), key=lambda kv: -kv[1])), "sum", round(sum(st.values()), 3))
print([(r["width"], r["height"]) for r in loop.rois])
