#!/usr/bin/env python3
"""images/sec of the conditional WGAN-GP loop iteration (conditional_gan/mnist/mnist_wgan_conditional.py:133-168) at the
reference's full width (1024) — BASELINE config 3 (global batch 1024 = 256 per GPU on 4 MI355X).

  python scripts/bench_wgan.py                 one GPU, batch 256
  python scripts/bench_wgan.py --gpus 4        starts 4 ranks itself (or run it under torch.distributed.run)

A "step" is one loop iteration: the critic update (every iteration) + 1/n_critic of a generator update (the reference runs it
every n_critic-th iteration, :157); both are timed separately and combined with that weight.  One JSON line, same contract as
bench.py.  Nominal MACs per image (SURVEY.md §8d convention: positions x k^2 x Cin x Cout): critic forward 70.2 M, generator
forward 195.9 M.  Critic update = 3 critic forwards + 1 generator forward + first-order backward of the real and fake passes
(2 forward-equivalents each, less the layer-1 dgrad) + the interpolate pass's gradient sweep (1) + backward-of-backward (conv_fwd +
wgrad up, wgrad + dgrad down: 4) ~ 12 x 70.2 + 195.9 = 1038 M MACs = 2.08 GFLOP.  Generator update = 3 x 195.9 + 2 x 70.2 = 728 M MACs."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib as BL  # noqa: E402
import torch  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--width", type=int, default=1024)
    BL.add_common_args(ap, steps=20, warmup=3)
    args = ap.parse_args(argv)
    R = BL.Ranks(args, os.path.abspath(__file__))
    from pcgan_amd import wgan as W, ops
    dev, dp = R.dev, R.dp
    hp = W.Hyperparameter(critic_size=args.width, generator_size=args.width, critic_hidden_size=args.width, batchsize=args.batch)
    critic, generator = W.build(dev, hp)
    c_opt, g_opt = W.make_optimizers(critic, generator)
    R.broadcast([critic, generator])
    rng = ops.DeviceRNG(1 + R.rank)
    B = args.batch
    x = rng.rand((B, 1, 28, 28), dev).mul_(2.0).sub_(1.0)
    lab = ops.onehot(rng.randint(0, 10, B, dev), 10)
    records = []

    gs = None if args.eager else W.GraphedSteps(critic, generator, c_opt, g_opt, hp, B, dev, dp=dp)

    # the draws of an iteration (noise :141, alpha :146, fake labels :161, noise :162) are launched eagerly in front of each replay
    def cstep(i, eager=False):
        noise, alpha = rng.randn((B, hp.latent_size), dev), rng.rand((B, 1), dev)
        if gs is None or eager:
            return W.critic_step(critic, generator, c_opt, hp, x, lab, noise, alpha, dp=dp)
        return gs.critic_step(x, lab, noise, alpha)

    def gstep(i, eager=False):
        fake, noise = ops.onehot(rng.randint(0, 10, B, dev), 10), rng.randn((B, hp.latent_size), dev)
        if gs is None or eager:
            return W.generator_step(critic, generator, g_opt, fake, noise, dp=dp)
        return gs.generator_step(fake, noise)

    def measure(fn):
        if gs is not None and args.warmup > 0:
            fn(0, eager=True)       # untimed: fills this stream's allocator pool for the event-sampled eager update (see bench_countergan.py)
        for i in range(args.warmup):
            fn(i)
        mid = args.steps // 2

        def one(i):                                       # one timed update runs eagerly with HIP events around every conv launch
            ops.set_conv_hook((lambda *r: records.append(r)) if i == mid else None)
            return fn(i, eager=(i == mid))
        dt, out = R.timed(one, args.steps)
        ops.set_conv_hook(None)
        return dt / args.steps, out

    tc, oc = measure(cstep)
    tg, og = measure(gstep)
    same = R.replicas_identical([critic, generator])
    scale = (args.width / 1024.0) ** 2
    fc, fg = 2 * 1038e6 * scale * B, 2 * 728e6 * scale * B
    it = tc + tg / hp.n_critic
    flops_it = fc + fg / hp.n_critic
    losses = {"critic_loss": float(oc["critic_loss"].item()), "gradient_penalty": float(oc["gradient_penalty"].item()),
              "generator_loss": float(og["generator_loss"].item())}
    if not all(v == v and abs(v) < 1e6 for v in losses.values()):
        sys.exit(f"non-finite losses: {losses}")
    cpu = None
    if R.rank == 0 and R.world == 1 and not args.no_cpu_baseline:
        from oracle import wgan_ref as WR                 # the checker's restatement: CPU-baseline leg only
        cb = B                                            # AT THE GPU BATCH (r02 timed 32 images against the GPU's 256), full width
        ohp = WR.Hyperparameter()
        ohp.critic_size = ohp.generator_size = ohp.critic_hidden_size = args.width
        ocr, og_ = WR.build(ohp, seed=1)
        oc_opt, og_opt = WR.make_optimizers(ocr, og_)
        xb, yb, zb, ab, y2, z2 = WR.synthetic_batch(ohp, cb, seed=0)
        ns, nw = (1, 0) if cb > 64 else (3, 1)            # a bounded sample: one update of each kind at the full batch
        mc, thr, avail = BL.cpu_median(lambda: WR.critic_step(ocr, og_, oc_opt, ohp, xb, yb, zb, ab), steps=ns, warmup=nw, threads=args.cpu_threads)
        mg, _, _ = BL.cpu_median(lambda: WR.generator_step(ocr, og_, og_opt, y2, z2), steps=ns, warmup=nw, threads=args.cpu_threads)
        cpu = {"value": round(cb / (mc + mg / hp.n_critic), 2), "unit": "images/sec", "cores": thr, "kind": "port", "cpu_model": BL.cpu_model(),
               "host_cpus_visible": avail,
               "sample": f"{ns} critic update(s) + {ns} generator update(s) at batch {cb} (= the GPU run's batch per GPU), width {args.width}, "
                         f"{nw} warm-up each, combined with the 1/n_critic weight; PyTorch-CPU fp32 restatement of mnist_wgan_conditional.py:133-168"}
    if R.rank == 0:
        roof = BL.conv_family_roofline(records, it, 1, flops_it, **dict(zip(("traffic", "traffic_file"), BL.pmc_traffic("wgan"))))
        if roof:
            roof["gemm_time_share"] = None     # events sample one critic and one generator update, not one weighted iteration
        BL.emit({
            "metric": "images/sec (loop iteration: critic update + 1/n_critic generator update) conditional WGAN-GP/mnist; % MFMA roofline",
            "value": round(R.world * B / it, 1), "unit": "images/sec", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(it * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"conditional_gan/mnist WGAN-GP (Generator / Critic width {args.width}, gradient penalty via hand-derived "
                                   f"backward of the backward), 28x28, batch {B} per GPU, n_critic {hp.n_critic}",
                       "global_batch": R.world * B, "parallelism": f"dp{R.world}"},
            "critic_update_ms": round(tc * 1e3, 3), "generator_update_ms": round(tg * 1e3, 3),
            "roofline": roof, "cpu_baseline": cpu, "final_losses": losses,
            "rccl_ranks": None if dp is None else dp.rccl_ranks(), "replicas_identical": same,
            "launch": "eager" if gs is None else "hip-graph replay (one graph per update); 1 of the timed updates of each kind eager with HIP events",
        })
    R.finish()


if __name__ == "__main__":
    main()
