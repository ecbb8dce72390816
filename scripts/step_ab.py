#!/usr/bin/env python3
"""Whole-step A/B of a tuning switch in ONE process, interleaved rounds (diagnostic; not the bench contract).

  python scripts/step_ab.py --model dcgan --ab edge_prio=0,2 [--rounds 6 --steps 10]
  python scripts/step_ab.py --model countergan --ab edge_prio=0,2
  python scripts/step_ab.py --model dcgan --ab pair=0,1          (pair: dcgan.train_step(pair=...) — the batched real + fake D pass)
  python scripts/step_ab.py --model dcgan --ab fullbn=0,1        (SequentialConvNet.fuse_full_window_bn)
  python scripts/step_ab.py --model dcgan --ab slabdefer=0,1     (SequentialConvNet.defer_slab_reductions)
  python scripts/step_ab.py --model dcgan --ab foldthin=0,1      (SequentialConvNet.fold_bn_apply_thin)

Per variant the step is captured as its own HIP graph (kernel arguments, tuning included, are baked in at capture), then the graphs
are replayed in alternating rounds; prints min / median ms per step of every variant.  Boxes differ by up to 20 % and drift within a
job, so only numbers from the same process are comparable."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import pcgan_amd  # noqa: E402
from pcgan_amd import ops  # noqa: E402
from pcgan_amd.nn import GraphedStep  # noqa: E402


def build_dcgan(dev, batch, variant):
    from pcgan_amd import dcgan as D
    torch.manual_seed(1)
    netG, netD = D.build(None, device="cpu")
    netG.to(dev); netD.to(dev)
    crit, optD, optG = D.make_optimizers(netG, netD)
    g = torch.Generator().manual_seed(1234)
    real = (torch.rand(batch, 1, 64, 64, generator=g) * 2 - 1).to(dev)
    noise = torch.randn(batch, 100, 1, 1, generator=g).to(dev)
    kw = dict(variant.get("kwargs", {}))
    return GraphedStep(lambda: D.train_step(netG, netD, crit, optD, optG, real, noise, **kw), {"real": real, "noise": noise}, [netG, netD], [optD, optG])   # (keeps the nets alive: nn.GraphedStep._keepalive)


def build_countergan(dev, batch, variant):
    from pcgan_amd import countergan as K
    torch.manual_seed(0)
    G, Dn, C = K.ResidualGenerator().to(dev), K.Discriminator().to(dev), K.CNNClassifier().to(dev)
    C.eval()
    for p in C.parameters():
        p.requires_grad = False
    opt_g, opt_d, bce, ce = K.make_optimizers(G, Dn)
    rng = ops.DeviceRNG(seed=1234)
    cfg = K.Config
    x = rng.rand((batch, 1, 28, 28), dev).mul_(2.0).sub_(1.0)
    y, t = rng.randint(0, cfg.num_classes, batch, dev), rng.randint(0, cfg.num_classes, batch, dev)
    m = rng.patch_mask(batch, 28, 28, cfg.patch_size, cfg.num_modifiable_patches, dev)
    return GraphedStep(lambda: K.train_step(G, Dn, C, opt_g, opt_d, bce, ce, x, y, t, m), {"x": x}, [G, Dn], [opt_g, opt_d])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dcgan", choices=["dcgan", "countergan"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--ab", required=True, help="switch=v0,v1[,v2...]: a pcg_tune_set switch, or pair=0,1 (dcgan.train_step keyword)")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--steps", type=int, default=10)
    a = ap.parse_args()
    pcgan_amd.load()
    dev = torch.device("cuda:0")
    key, vals = a.ab.split("=")
    vals = [int(v) for v in vals.split(",")]
    batch = a.batch or (512 if a.model == "dcgan" else 1024)
    graphs = {}
    for v in vals:
        variant = {}
        if key == "pair":
            variant["kwargs"] = {"pair": bool(v)}
        elif key == "overlap":           # nn.SequentialConvNet.wgrad_overlap (class switch: baked into the graph at capture)
            from pcgan_amd.nn import SequentialConvNet
            SequentialConvNet.wgrad_overlap = "bn" if v else None
        elif key == "fullbn":            # nn.SequentialConvNet.fuse_full_window_bn: D5's grad-input through D4's BatchNorm backward unwritten
            from pcgan_amd.nn import SequentialConvNet
            SequentialConvNet.fuse_full_window_bn = bool(v)
        elif key == "foldthin":          # nn.SequentialConvNet.fold_bn_apply_thin: G4's BatchNorm + ReLU inside G5's loads
            from pcgan_amd.nn import SequentialConvNet
            SequentialConvNet.fold_bn_apply_thin = bool(v)
        elif key == "slabdefer":         # nn.SequentialConvNet.defer_slab_reductions: one slab reduction per backward sweep
            from pcgan_amd.nn import SequentialConvNet
            SequentialConvNet.defer_slab_reductions = bool(v)
        else:
            ops.tune(key, v)
        graphs[v] = (build_dcgan if a.model == "dcgan" else build_countergan)(dev, batch, variant)
        if key == "overlap":
            from pcgan_amd.nn import SequentialConvNet
            SequentialConvNet.wgrad_overlap = None
        elif key == "fullbn":
            from pcgan_amd.nn import SequentialConvNet
            SequentialConvNet.fuse_full_window_bn = True
        elif key == "slabdefer":
            from pcgan_amd.nn import SequentialConvNet
            SequentialConvNet.defer_slab_reductions = False
        elif key == "foldthin":
            from pcgan_amd.nn import SequentialConvNet
            SequentialConvNet.fold_bn_apply_thin = True
        elif key != "pair":
            ops.tune(key, -1)
    res = {v: [] for v in vals}
    for v in vals:
        for _ in range(3):
            graphs[v].replay()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v in vals:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                graphs[v].replay()
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / a.steps)
    line = f"{a.model} batch {batch}"
    for v in vals:
        r = sorted(res[v])
        line += f" | {key}={v}: med {r[len(r) // 2]:8.4f} ms min {r[0]:8.4f}"
    print(line, flush=True)


if __name__ == "__main__":
    main()
