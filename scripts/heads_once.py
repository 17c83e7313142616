"""One forward + backward of the CLIP / instance heads on a 960x540 feature image (for counter passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pogs_amd.heads import FieldHeads, HeadNames  # noqa: E402

P, dev = 960 * 540, torch.device("cuda:0")
heads = FieldHeads(device=dev, seed=0)
feats = torch.randn(P, 96, device=dev).requires_grad_(True)
scale = torch.rand(P, 1, device=dev)
for _ in range(2):
    o = heads.get_outputs_from_feature(feats, scale)
    (o[HeadNames.CLIP].sum() + o[HeadNames.INSTANCE].sum()).backward()
torch.cuda.synchronize()
