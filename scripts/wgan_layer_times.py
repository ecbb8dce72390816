#!/usr/bin/env python3
"""Per-layer timing of the conv family inside the WGAN-GP updates (width 1024, batch 256): every MFMA conv launch of one eager
critic update and one eager generator update, bracketed by HIP events and keyed by (kernel, op, geometry).  Diagnostic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pcgan_amd  # noqa: E402
from pcgan_amd import ops, wgan as W  # noqa: E402


def main():
    pcgan_amd.load()
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    hp = W.Hyperparameter(critic_size=1024, generator_size=1024, critic_hidden_size=1024, batchsize=B)
    critic, generator = W.build(dev, hp)
    c_opt, g_opt = W.make_optimizers(critic, generator)
    rng = ops.DeviceRNG(1)
    x = rng.rand((B, 1, 28, 28), dev).mul_(2.0).sub_(1.0)
    lab = ops.onehot(rng.randint(0, 10, B, dev), 10)
    real_label = ops._conv_label
    ops._conv_label = lambda g, op: f"{real_label(g, op)} | {op} B{g.B} {g.Cin}->{g.Cout} {g.IH}x{g.IW}->{g.OH}x{g.OW} k{g.KH}s{g.stride}p{g.pad}"
    for which in ("critic", "generator"):
        rec = []

        def run():
            if which == "critic":
                W.critic_step(critic, generator, c_opt, hp, x, lab, rng.randn((B, hp.latent_size), dev), rng.rand((B, 1), dev))
            else:
                W.generator_step(critic, generator, g_opt, ops.onehot(rng.randint(0, 10, B, dev), 10), rng.randn((B, hp.latent_size), dev))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ops.set_conv_hook(lambda label, flops, e0, e1: rec.append((label, flops, e0, e1)))
        reps = 5
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(reps):
            run()
        t1.record()
        torch.cuda.synchronize()
        ops.set_conv_hook(None)
        agg = {}
        order = []
        for label, flops, e0, e1 in rec:
            if label not in agg:
                agg[label] = [0, 0.0, 0.0]; order.append(label)
            a = agg[label]; a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1)
        tot_ms = sum(a[2] for a in agg.values()) / reps
        print(f"== {which} update: eager {t0.elapsed_time(t1) / reps:.3f} ms; conv family {tot_ms:.3f} ms, "
              f"{sum(a[1] for a in agg.values()) / sum(a[2] for a in agg.values()) / 1e9:.1f} TFLOP/s")
        for label in order:
            n, fl, ms = agg[label]
            print(f"  {n / reps:4.1f}x {ms / n * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF  {label}")


if __name__ == "__main__":
    main()
