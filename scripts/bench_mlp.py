"""Time the fused channel MLP against the torch operators it replaces, on a 1080p feature image.

Usage: python scripts/bench_mlp.py [--pixels N] [--reps R]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pogs_amd.mlp import FeatureMLP  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
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
    ap.add_argument("--pixels", type=int, default=1920 * 1080)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    P = args.pixels
    dev = torch.device("cuda:0")
    m = FeatureMLP(device=dev)
    for p in m.parameters():
        p.requires_grad_(False)
    x = torch.randn(1, P, 64, device=dev)
    al = torch.rand(1, P, 1, device=dev)
    v = torch.randn(1, P, 64, device=dev)

    def torch_fwd(xx):
        h = torch.where(al > 0, xx / al.clamp(min=1e-6), torch.zeros(64, device=dev)).view(-1, 64)
        return torch.nn.Sequential.forward(m, h).view(1, P, 64)

    def fused_fwd_bwd():
        xx = x.detach().requires_grad_(True)
        m(xx, al).backward(v)

    def torch_fwd_bwd():
        xx = x.detach().requires_grad_(True)
        torch_fwd(xx).backward(v)

    with torch.no_grad():
        t_f = timed(lambda: m(x, al), args.reps)
        t_t = timed(lambda: torch_fwd(x), args.reps)
    t_fb = timed(fused_fwd_bwd, args.reps)
    t_tb = timed(torch_fwd_bwd, args.reps)
    for p in m.parameters():
        p.requires_grad_(True)

    def clear():
        for p in m.parameters():
            p.grad = None

    t_fw = timed(lambda: (clear(), fused_fwd_bwd()), args.reps)
    t_tw = timed(lambda: (clear(), torch_fwd_bwd()), args.reps)
    flop = 4 * 64 * 64 * 2 * P
    print(f"pixels {P}")
    print(f"forward        fused {t_f:.3f} ms ({flop / t_f / 1e9:.1f} TFLOP/s, {P * 516 / t_f / 1e6:.0f} GB/s)   "
          f"torch {t_t:.3f} ms   x{t_t / t_f:.2f}")
    print(f"fwd + d/dfeats fused {t_fb:.3f} ms   torch {t_tb:.3f} ms   x{t_tb / t_fb:.2f}")
    print(f"fwd + d/dfeats + d/dweights (training) fused {t_fw:.3f} ms   torch {t_tw:.3f} ms   x{t_tw / t_fw:.2f}")


if __name__ == "__main__":
    main()
