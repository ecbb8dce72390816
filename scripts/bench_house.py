#!/usr/bin/env python3
"""Step time of the tabular CounteRGAN training step (house_sales_kc_usa/trainer.py:241-316) on one MI355X — secondary
measurement; the contract bench is bench.py (DCGAN).  Two modes: eager (one host launch per
kernel) and --graph (the whole step — G fwd, D step, G step, both Adams — captured once in a HIP graph and replayed; the
per-step draws are written into static buffers by three RNG launches outside the graph)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgan_amd
from pcgan_amd import house as H, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--warmup", type=int, default=20)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--cpu-baseline", action="store_true", help="also time oracle/house_ref.py (PyTorch CPU) on the same batch size")
args = ap.parse_args()
dev = torch.device("cuda:0")
G, D, C = H.build(dev, seed=0)
opt_g, opt_d = H.make_optimizers(G, D)
norm = H.cat_norm_maps(G, H.CONFIG, dev)
rng = ops.DeviceRNG(1)
B = args.batch
x = torch.rand(B, 17, device=dev)
y = torch.randint(0, 4, (B,), device=dev)
t, mask, noise = H.draw_batch_randoms(rng, G, y, H.CONFIG, dev)

def draws():
    t2, m2, n2 = H.draw_batch_randoms(rng, G, y, H.CONFIG, dev)
    t.copy_(t2); mask.copy_(m2); noise.copy_(n2)

def step():
    return H.train_step(G, D, C, opt_g, opt_d, x, y, t, mask, norm, gumbel=noise)

if args.graph:
    gs = H.GraphedTrainStep(G, D, C, opt_g, opt_d, norm, B)
    gs.x.copy_(x); gs.y.copy_(y)
    t, mask, noise = gs.target_y, gs.mask, gs.noise
    def run():
        global out
        draws(); out = gs.replay()
else:
    def run():
        global out
        draws(); out = step()
for _ in range(args.warmup):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
line = (f"house-sales counteRGAN batch {B} {'graph' if args.graph else 'eager'}: {dt * 1e3:.3f} ms/step  {B / dt:.0f} rows/s  "
        f"D_loss {out['D_loss'].item():.4f} G_loss {out['G_loss'].item():.4f}")
if args.cpu_baseline:
    from oracle import house_ref as HR
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    oG, oD, oC = HR.build(0)
    o_g, o_d = HR.make_optimizers(oG, oD)
    xb, yb, tb, mb, gb = HR.synthetic_batch(B, 0)
    nm = HR.cat_norm_maps()
    for _ in range(3):
        HR.house_step(oG, oD, oC, o_g, o_d, xb, yb, tb, mb, gb, nm)
    n = max(5, min(100, int(2.0 / max(dt, 1e-4) / 50)))
    t0 = time.perf_counter()
    for _ in range(n):
        HR.house_step(oG, oD, oC, o_g, o_d, xb, yb, tb, mb, gb, nm)
    ct = (time.perf_counter() - t0) / n
    line += f"  | oracle (PyTorch CPU, {torch.get_num_threads()} threads): {ct * 1e3:.3f} ms/step  {B / ct:.0f} rows/s"
print(line)
