"""The tracker's default loop alone (bench.py --config cfg3's scene): 1 warm-up frame + FRAMES frames of 50 iterations,
for a kernel trace (rocprofv3 --kernel-trace --stats -- python3 scripts/profile_tracker.py)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pogs_amd.tracking import TrackingLoop  # noqa: E402

FRAMES = int(os.environ.get("FRAMES", "3"))
dev = torch.device("cuda:0")
loop = TrackingLoop.synthetic(dev, n=200_000, width=640, height=360, groups=4, seed=0)
loop.frame(50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(FRAMES):
    loop.frame(50)
torch.cuda.synchronize()
print("ms/frame", (time.perf_counter() - t0) / FRAMES * 1e3, flush=True)
