#!/usr/bin/env python3
"""Time the fused thin grad-input + BatchNorm backward (pcg_conv2d_fwd_bnbwd_thin) at DCGAN's G5 / G4 shape: HIP events around N calls.
Run under rocprofv3 --kernel-trace --stats for the two passes separately."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pcgan_amd as pcg  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
pcg.load()
ops = pcg.ops
d = torch.device("cuda:0")
B, C = a.batch, 64
g = ops.conv_geom(B, 64, 64, 1, C, 4, 4, 2, 1)          # the ConvTranspose2d(64, 1, 4, 2, 1) of mnist_dcgan.py:88 as a conv geometry
gen = torch.Generator().manual_seed(3)
x = torch.randn(B, 64, 64, 1, generator=gen).to(d)
w = (torch.randn(C, 4, 4, 1, generator=gen) * 0.1).to(d)
z = torch.randn(B, 32, 32, C, generator=gen).to(d)
mean, invstd = z.mean((0, 1, 2)).contiguous(), (1.0 / z.std((0, 1, 2))).contiguous()
gamma, beta = torch.ones(C, device=d), torch.zeros(C, device=d)
dg, db = torch.zeros(C, device=d), torch.zeros(C, device=d)
for _ in range(3):
    ops.thin_fwd_bn_bwd(g, x, w, z, mean, invstd, gamma, beta, ops.ACT_RELU, 0.0, dg, db, False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(a.iters):
    ops.thin_fwd_bn_bwd(g, x, w, z, mean, invstd, gamma, beta, ops.ACT_RELU, 0.0, dg, db, False)
e1.record()
torch.cuda.synchronize()
print(f"thin grad-input + BatchNorm backward, B={B}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per call (sums + finalize + apply)")
