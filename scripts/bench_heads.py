"""Time the CLIP / instance heads against the torch operators (float32 and float16 GEMMs) on a half-resolution
1080p feature image (960 x 540 rows of 96 features), forward and forward+backward with weight gradients.

Usage: python scripts/bench_heads.py [--rows N] [--reps R]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pogs_amd.heads import FieldHeads, HeadNames  # noqa: E402


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=960 * 540)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    P, dev = args.rows, torch.device("cuda:0")
    heads = FieldHeads(device=dev, seed=0)
    feats = torch.randn(P, 96, device=dev)
    scale = torch.rand(P, 1, device=dev)
    vc, vi = torch.randn(P, 512, device=dev), torch.randn(P, 128, device=dev)

    def torch_heads(f, dtype):
        outs = []
        for net, x, eps in ((heads.clip_net, torch.cat([f, scale], -1), 0.0), (heads.instance_net, f, 1e-5)):
            h = net.padded_input(x).to(dtype)
            ms = net.matrices()
            for i, w in enumerate(ms):
                h = h @ w.to(dtype).t()
                if i < len(ms) - 1:
                    h = torch.relu(h)
            h = h.float()
            outs.append(h / (h.norm(dim=-1, keepdim=True) + eps))
        return outs

    def fused(f):
        o = heads.get_outputs_from_feature(f, scale)
        return o[HeadNames.CLIP], o[HeadNames.INSTANCE]

    def train(fn):
        for p in heads.parameters():
            p.grad = None
        f = feats.detach().requires_grad_(True)
        c, i = fn(f)
        ((c * vc).sum() + (i * vi).sum()).backward()

    flop = 2 * P * sum(n * k for net in (heads.clip_net, heads.instance_net) for n, k in net.shapes)
    with torch.no_grad():
        t_f = timed(lambda: fused(feats), args.reps)
        t_32 = timed(lambda: torch_heads(feats, torch.float32), args.reps)
        t_16 = timed(lambda: torch_heads(feats, torch.float16), args.reps)
    b_f = timed(lambda: train(fused), args.reps)
    b_32 = timed(lambda: train(lambda f: torch_heads(f, torch.float32)), args.reps)
    b_16 = timed(lambda: train(lambda f: torch_heads(f, torch.float16)), args.reps)
    print(f"rows {P}, {flop / 1e9:.1f} GFLOP forward")
    print(f"forward            fused {t_f:.3f} ms ({flop / t_f / 1e9:.0f} TFLOP/s)   torch f32 {t_32:.3f} ms   torch f16 {t_16:.3f} ms")
    print(f"forward + backward fused {b_f:.3f} ms   torch f32 {b_32:.3f} ms   torch f16 {b_16:.3f} ms")


if __name__ == "__main__":
    main()
