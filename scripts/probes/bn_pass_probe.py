#!/usr/bin/env python3
"""Time the BatchNorm apply / backward passes alone at the bench shapes (HIP events around N calls): rows x C of the DCGAN and CounteRGAN
activations.  A/B knobs of the streaming kernels come from the environment (PCG_BN_BLOCKS, PCG_BN_DEPTH, PCG_BN_NT)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcgan_amd as pcg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
pcg.load()
ops = pcg.ops
d = torch.device("cuda:0")
SHAPES = [("DCGAN G4 / D1 512x32x32", 512 * 32 * 32, 64), ("DCGAN D2 pair 1024x16x16", 1024 * 16 * 16, 128), ("DCGAN D3 512x8x8", 512 * 8 * 8, 256),
          ("CounteRGAN 1024x28x28", 1024 * 28 * 28, 64)]


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters * 1e3


for name, rows, C in SHAPES:
    x = torch.randn(rows, C, device=d)
    dy = torch.randn(rows, C, device=d)
    mean, invstd = x.mean(0).contiguous(), (1.0 / x.std(0)).contiguous()
    gamma, beta = torch.ones(C, device=d), torch.zeros(C, device=d)
    dg, db = torch.zeros(C, device=d), torch.zeros(C, device=d)
    y = torch.empty_like(x)
    mb = rows * C * 4 / 1e6
    t = timed(lambda: ops.bn_apply_act(x, C, mean, invstd, gamma, beta, ops.ACT_LRELU, 0.2, out=y))
    print(f"{name:28s} apply     {t:7.1f} us  {2 * mb / t:5.2f} TB/s (read + write)")
    t = timed(lambda: ops.bn_act_bwd(dy, x, None, C, mean, invstd, gamma, ops.ACT_LRELU, 0.2, dg, db, False, beta=beta))
    print(f"{name:28s} backward  {t:7.1f} us  (column sums + finalize + apply: 5 tensor passes = {5 * mb / t:5.2f} TB/s)")
