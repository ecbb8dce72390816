#!/usr/bin/env python3
"""Step time of the conditional WGAN-GP loop iteration (mnist_wgan_conditional.py:133-168) on one MI355X at the reference's
full width (1024) — secondary measurement (BASELINE config: global batch 1024 = 256 per GPU on 4);
the contract bench is bench.py (DCGAN).  Reports the critic update (every iteration) and the generator update (every n_critic-th)
separately and their n_critic-weighted mean.

Nominal MACs per image (SURVEY.md section 8d convention: positions x k^2 x Cin x Cout): critic forward 70.2 M, generator forward
195.9 M.  Critic update = 3 critic forwards + 1 generator forward + first-order backward of the real and fake passes (2 forward-
equivalents each, less the layer-1 dgrad) + the interpolate pass's gradient sweep (1) + backward-of-backward (conv_fwd + wgrad up,
wgrad + dgrad down: 4) ~ 12 x 70.2 + 195.9 = 1038 M MACs = 2.08 GFLOP.  Generator update = 3 x 195.9 + 2 x 70.2 = 728 M MACs."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgan_amd
from pcgan_amd import wgan as W, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--width", type=int, default=1024)
args = ap.parse_args()
dev = torch.device("cuda:0")
hp = W.Hyperparameter(critic_size=args.width, generator_size=args.width, critic_hidden_size=args.width, batchsize=args.batch)
critic, generator = W.build(dev, hp)
c_opt, g_opt = W.make_optimizers(critic, generator)
rng = ops.DeviceRNG(1)
B = args.batch
x = rng.rand((B, 1, 28, 28), dev) * 2 - 1
lab = ops.onehot(rng.randint(0, 10, B, dev), 10)

def cstep():
    return W.critic_step(critic, generator, c_opt, hp, x, lab, rng.randn((B, hp.latent_size), dev), rng.rand((B, 1), dev))
def gstep():
    return W.generator_step(critic, generator, g_opt, ops.onehot(rng.randint(0, 10, B, dev), 10), rng.randn((B, hp.latent_size), dev))
def timeit(fn):
    for _ in range(args.warmup):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps, out
tc, oc = timeit(cstep)
tg, og = timeit(gstep)
scale = (args.width / 1024.0) ** 2
fc, fg = 2 * 1038e6 * scale * B, 2 * 728e6 * scale * B
it = tc + tg / hp.n_critic
print(f"WGAN-GP width {args.width} batch {B}: critic update {tc * 1e3:.2f} ms ({fc / tc / 1e12:.1f} TFLOP/s nominal), generator update "
      f"{tg * 1e3:.2f} ms ({fg / tg / 1e12:.1f} TFLOP/s), loop iteration (n_critic={hp.n_critic}) {it * 1e3:.2f} ms = {B / it:.0f} img/s "
      f"({(fc + fg / hp.n_critic) / it / 1e12:.1f} TFLOP/s = {(fc + fg / hp.n_critic) / it / 157.3e12 * 100:.1f}% of fp32-MFMA peak)  "
      f"critic_loss {oc['critic_loss'].item():.4f} gp {oc['gradient_penalty'].item():.4f} g_loss {og['generator_loss'].item():.4f}")
