#!/usr/bin/env python3
"""images/sec of the CounteRGAN/mnist training step (conditional_counteRGAN/mnist/trainer.py:89-123) — BASELINE config 4
(batch 1024 per GPU, data-parallel over up to 8 MI355X; `--batch 128 --gpus 8` is the strong-scaling shard of a 1024 global batch).

  python scripts/bench_countergan.py                       one GPU, batch 1024, HIP-graph replay
  python scripts/bench_countergan.py --gpus 8              starts 8 ranks itself (or run it under torch.distributed.run)

One JSON line, same contract as bench.py: `roofline` = the fp32-MFMA conv family measured with HIP events on event-sampled eager
steps (13 of the step's convolutions are 3x3 64->64 at 28x28: 91 % of its FLOPs); `cpu_baseline` = oracle/countergan_ref.py
(PyTorch-CPU restatement of the loop body) on a bounded sample."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib as BL  # noqa: E402
import torch  # noqa: E402

ALGO_GFLOP_PER_IMAGE = 2.476


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU")
    BL.add_common_args(ap, steps=10, warmup=3)
    args = ap.parse_args(argv)
    R = BL.Ranks(args, os.path.abspath(__file__))
    from pcgan_amd import countergan as K, ops
    dev, dp = R.dev, R.dp
    torch.manual_seed(0)
    G, D, C = K.ResidualGenerator().to(dev), K.Discriminator().to(dev), K.CNNClassifier().to(dev)
    C.eval()
    for p in C.parameters():
        p.requires_grad = False
    opt_g, opt_d, bce, ce = K.make_optimizers(G, D)
    R.broadcast([G, D])
    # synthetic MNIST-shaped shards drawn on the device: x ~ U[-1,1), labels / targets ~ U{0..9}, masks =
    # 10 of 16 7x7 patches; each rank its own stream
    rng = ops.DeviceRNG(seed=1234 + R.rank)
    cfg = K.Config

    def synth():
        x = rng.rand((args.batch, 1, 28, 28), dev).mul_(2.0).sub_(1.0)           # setup, not the step
        return (x, rng.randint(0, cfg.num_classes, args.batch, dev), rng.randint(0, cfg.num_classes, args.batch, dev),
                rng.patch_mask(args.batch, 28, 28, cfg.patch_size, cfg.num_modifiable_patches, dev))
    batches = [synth() for _ in range(2)]

    def eager(i):
        x, y, t, m = batches[i % 2]
        return K.train_step(G, D, C, opt_g, opt_d, bce, ce, x, y, t, m, dp=dp)

    gs = None
    if not args.eager:
        from pcgan_amd.nn import GraphedStep
        x, y, t, m = (b.clone() for b in batches[0])
        if dp is None:
            gs = GraphedStep(lambda: K.train_step(G, D, C, opt_g, opt_d, bce, ce, x, y, t, m), {"x": x, "y": y, "t": t, "m": m}, [G, D], [opt_g, opt_d])
        else:
            gs = GraphedStep(lambda d: K.train_step(G, D, C, opt_g, opt_d, bce, ce, x, y, t, m, dp=d), {"x": x, "y": y, "t": t, "m": m},
                             [G, D], [opt_g, opt_d], dp=dp)

    records = []
    sampled = {args.steps // 2} if args.steps > 1 else {0}

    def step(i):
        if gs is None or i in sampled_now:
            return eager(i)
        bx, by, bt, bm = batches[i % 2]
        gs.load(x=bx, y=by, t=bt, m=bm)
        return gs.replay()

    sampled_now = set()
    if gs is not None and args.warmup > 0:
        eager(0)       # untimed: the event-sampled timed step runs eagerly on THIS stream — its activations come out of the caching allocator's
                       # pool of this stream, which only an eager step fills (r04: a cold pool cost one timed step +10 ms of hipMalloc)
    for i in range(args.warmup):
        step(i)
    sampled_now = sampled

    def timed_step(i):
        ops.set_conv_hook((lambda *r: records.append(r)) if i in sampled else None)
        return step(i)

    dt, out = R.timed(timed_step, args.steps)
    ops.set_conv_hook(None)
    sec = dt / args.steps
    same = R.replicas_identical([G, D])
    losses = {k: float(out[k].item()) for k in ("g_loss", "d_loss")}
    if not all(v == v and abs(v) < 1e4 for v in losses.values()):
        sys.exit(f"non-finite losses: {losses}")
    cpu = None
    if R.rank == 0 and R.world == 1 and not args.no_cpu_baseline:
        from oracle import countergan_ref as CR        # the checker's restatement: CPU-baseline leg only
        cb = args.batch                                # AT THE GPU BATCH (r02 timed 128 against the GPU's 1024): ~10-36 s per CPU step at 1024
        oG, oD, oC = CR.build(seed=0)
        o = CR.make_optimizers(oG, oD)
        xb, yb, tb, mb = CR.synthetic_batch(cb, seed=0)
        med, thr, avail = BL.cpu_median(lambda: CR.countergan_step(oG, oD, oC, *o, xb, yb, tb, mb), steps=1 if cb > 256 else 3, warmup=0 if cb > 256 else 1,
                                        threads=args.cpu_threads)
        cpu = {"value": round(cb / med, 2), "unit": "images/sec", "cores": thr, "kind": "port", "cpu_model": BL.cpu_model(),
               "host_cpus_visible": avail,
               "sample": (f"ONE step at batch {cb} (= the GPU run's batch per GPU; a bounded sample, no warm-up: a CPU step takes 10-36 s)"
                          if cb > 256 else f"median of 3 steps at batch {cb} (= the GPU run's batch) after 1 warm-up")
                         + "; PyTorch-CPU fp32 restatement of trainer.py:96-123 (oracle/countergan_ref.py)"}
    if R.rank == 0:
        BL.emit({
            "metric": "images/sec (G+D step) CounteRGAN/mnist + frozen classifier; % MFMA roofline",
            "value": round(R.world * args.batch / sec, 1), "unit": "images/sec", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(sec * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"conditional_counteRGAN/mnist CounteRGAN (ResidualGenerator 6 blocks, Discriminator, frozen CNNClassifier), "
                                   f"28x28, batch {args.batch} per GPU, full step incl. BatchNorm, BCE/CE/L1 losses, Adam x2",
                       "global_batch": R.world * args.batch, "parallelism": f"dp{R.world}"},
            "roofline": BL.conv_family_roofline(records, sec, len(sampled), ALGO_GFLOP_PER_IMAGE * 1e9 * args.batch, **dict(zip(("traffic", "traffic_file"), BL.pmc_traffic("countergan")))),
            "cpu_baseline": cpu, "final_losses": losses,
            "rccl_ranks": None if dp is None else dp.rccl_ranks(), "replicas_identical": same,
            "launch": "eager" if gs is None else f"hip-graph replay ({len(gs.program)} segment(s)); {len(sampled)} of {args.steps} timed steps eager with HIP events",
        })
    R.finish()


if __name__ == "__main__":
    main()
