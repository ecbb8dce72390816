#!/usr/bin/env python3
"""Throughput of the CounteRGAN/mnist training step (trainer.py:89-123) on one MI355X — secondary measurement
(BASELINE config 4 at its per-GPU shard); the contract bench is bench.py (DCGAN)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgan_amd
from pcgan_amd import countergan as K
from oracle import countergan_ref as CR

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--graph", action="store_true", help="capture the step in a HIP graph (pcgan_amd.nn.GraphedStep)")
args = ap.parse_args()
dev = "cuda:0"
torch.manual_seed(0)
G, D, C = K.ResidualGenerator().to(dev), K.Discriminator().to(dev), K.CNNClassifier().to(dev)
C.eval()
for p in C.parameters():
    p.requires_grad = False
opt_g, opt_d, bce, ce = K.make_optimizers(G, D)
batches = [tuple(t.to(dev) for t in CR.synthetic_batch(args.batch, seed=s)) for s in range(2)]
def step(i):
    x, y, t, m = batches[i % 2]
    return K.train_step(G, D, C, opt_g, opt_d, bce, ce, x, y, t, m)
if args.graph:
    from pcgan_amd.nn import GraphedStep
    x, y, t, m = (b.clone() for b in batches[0])
    gs = GraphedStep(lambda: K.train_step(G, D, C, opt_g, opt_d, bce, ce, x, y, t, m), {"x": x, "y": y, "t": t, "m": m}, [G, D], [opt_g, opt_d])
    def step(i):
        bx, by, bt, bm = batches[i % 2]
        gs.load(x=bx, y=by, t=bt, m=bm)
        return gs.replay()
for i in range(args.warmup):
    out = step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(args.steps):
    out = step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
gflop = 2.476  # per image per step, SURVEY.md §8d
print(f"counteRGAN/mnist batch {args.batch}: {dt * 1e3:.2f} ms/step  {args.batch / dt:.0f} img/s  "
      f"{gflop * args.batch / dt / 1e3:.1f} TFLOP/s algorithmic ({gflop * args.batch / dt / 1e3 / 157.3 * 100:.1f}% of fp32-MFMA peak)  "
      f"g_loss {out['g_loss'].item():.4f} d_loss {out['d_loss'].item():.4f}")
