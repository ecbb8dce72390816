#!/usr/bin/env python3
"""Per-layer timing of the conv family inside the CounteRGAN/mnist step (batch 1024): every MFMA conv launch of eager steps,
bracketed by HIP events and keyed by (kernel, op, geometry).  Diagnostic (the WGAN-GP twin: scripts/wgan_layer_times.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pcgan_amd  # noqa: E402
from pcgan_amd import countergan as K, ops  # noqa: E402


def main():
    pcgan_amd.load()
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    torch.manual_seed(0)
    G, D, C = K.ResidualGenerator().to(dev), K.Discriminator().to(dev), K.CNNClassifier().to(dev)
    C.eval()
    for prm in C.parameters():
        prm.requires_grad = False
    opt_g, opt_d, bce, ce = K.make_optimizers(G, D)
    rng = ops.DeviceRNG(1)
    cfg = K.Config
    x = rng.rand((B, 1, 28, 28), dev).mul_(2.0).sub_(1.0)
    y, t = rng.randint(0, cfg.num_classes, B, dev), rng.randint(0, cfg.num_classes, B, dev)
    m = rng.patch_mask(B, 28, 28, cfg.patch_size, cfg.num_modifiable_patches, dev)
    real_label = ops._conv_label
    ops._conv_label = lambda g, op: f"{real_label(g, op)} | {op} B{g.B} {g.Cin}->{g.Cout} {g.IH}x{g.IW}->{g.OH}x{g.OW} k{g.KH}s{g.stride}p{g.pad}"
    run = lambda: K.train_step(G, D, C, opt_g, opt_d, bce, ce, x, y, t, m)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    rec = []
    ops.set_conv_hook(lambda label, flops, e0, e1: rec.append((label, flops, e0, e1)))
    reps = 3
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        run()
    t1.record()
    torch.cuda.synchronize()
    ops.set_conv_hook(None)
    agg, order = {}, []
    for label, flops, e0, e1 in rec:
        if label not in agg:
            agg[label] = [0, 0.0, 0.0]; order.append(label)
        a = agg[label]; a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1)
    tot_ms = sum(a[2] for a in agg.values()) / reps
    print(f"== step: eager {t0.elapsed_time(t1) / reps:.3f} ms; conv family {tot_ms:.3f} ms, "
          f"{sum(a[1] for a in agg.values()) / sum(a[2] for a in agg.values()) / 1e9:.1f} TFLOP/s")
    for label in order:
        n, fl, ms = agg[label]
        print(f"  {n / reps:4.1f}x {ms / n * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF  total {ms / reps:6.3f} ms  {label}")


if __name__ == "__main__":
    main()
