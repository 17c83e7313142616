"""Per-entry-point GPU time of the tracker's whole-frame loop, ms per iteration.  FUSE=1: one 68-channel render;
NORGB=1: first render without colours."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd import rendering  # noqa: E402
from pogs_amd.tracking import TrackingLoop  # noqa: E402

dev = torch.device("cuda:0")
loop = TrackingLoop.synthetic(dev, n=200_000, width=640, height=360, groups=4, seed=0)
loop.fuse_renders = os.environ.get("FUSE") == "1"
loop.skip_unused_rgb = os.environ.get("NORGB") == "1"
loop.frame(50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(4):
    loop.frame(50)
torch.cuda.synchronize()
print("ms/frame", (time.perf_counter() - t0) / 4 * 1e3, flush=True)
timer = rendering.StageTimer()
rendering.set_stage_timer(timer)
loop.frame(50)
torch.cuda.synchronize()
rendering.set_stage_timer(None)
st = {k: round(v[1] / 50, 4) for k, v in timer.summary().items()}
print(dict(sorted(st.items(), key=lambda kv: -kv[1])), "sum", round(sum(st.values()), 3))
