#!/usr/bin/env python3
"""rows/sec of the tabular CounteRGAN training step (conditional_counteRGAN/house_sales_kc_usa/trainer.py:241-316) on one MI355X —
BASELINE config 5 (batch 4096, single GPU: the path does not shard, "replicas only").

  python scripts/bench_house.py                  batch 4096, the whole step (G fwd, D step, G step, both Adams) replayed as ONE single-stream HIP graph
  python scripts/bench_house.py --overlap        A/B: the classifier term on a parallel graph branch
  python scripts/bench_house.py --eager          one host launch per kernel

One JSON line, same contract as bench.py.  The step is ~1 MFLOP per row on 17..256-wide layers: no MFMA claim.  `roofline` is
stated against HBM with `bound: "launch/hbm"`: achieved = algorithmic bytes per step (the batch's inputs and draws once; every
parameter read once per pass that uses it; Adam's 28 B per trained parameter; the per-row activations that backward needs, written
once and read once) / step time — the step is launch-latency-bound, the fraction is small by construction and `launches_per_step`
(from the committed rocprofv3 summary of this command) is the number that explains it."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _benchlib as BL  # noqa: E402
import torch  # noqa: E402


def through_trainer(args, R, H, ops, dev):
    import time
    import numpy as np
    B, N = args.batch, 17290                                # the reference's training split (80 % of 21,613 rows)
    rs = np.random.RandomState(0)
    X, y = rs.random_sample((N, H.CONFIG["input_dim"])).astype(np.float32), rs.randint(0, H.CONFIG["num_classes"], N)
    per_epoch = N // B
    e_short = 8
    e_long = e_short + max(1, -(-args.steps // per_epoch))

    def run(epochs):
        G, _, C = H.build(dev, seed=0)
        cfg = dict(H.CONFIG, epochs=epochs, batch_size=B, cuda=str(dev), seed=42)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hist = H.train_countergan(G, cfg, X, y, C, verbose=False, save=False)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, hist

    run(2)                                                   # code objects, allocator
    t_s, _ = run(e_short)
    t_l, hist = run(e_long)
    its = (e_long - e_short) * per_epoch
    sec = (t_l - t_s) / its
    if not all(np.isfinite(v) for k in ("d_losses", "g_losses", "pred_gain", "D_grad") for v in hist[k]):
        sys.exit("non-finite history")
    BL.emit({
        "metric": "rows/sec (G+D step) tabular CounteRGAN house_sales_kc_usa; through house.train_countergan", "value": round(B / sec, 1),
        "unit": "rows/sec", "n_gpus": 1, "steps": its, "warmup": e_short * per_epoch, "ms_per_step": round(sec * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"house.train_countergan(generator, config, X_train, y_train, clf_model): {N} rows, batch {B} ({per_epoch} iterations "
                               f"per epoch, drop_last), diagnostics + epoch summaries + D_grad variant; marginal time per iteration "
                               f"between {e_short} and {e_long} epochs", "global_batch": B, "parallelism": "dp1"},
        "roofline": None, "cpu_baseline": None,
        "epoch_tail": {k: hist[k][-1] for k in ("d_losses", "g_losses", "pred_gain", "sparsity", "l2_reg", "class_flip_rate", "G_grad", "D_grad")},
        "seconds": {"short": round(t_s, 4), "long": round(t_l, 4)},
    })
    R.finish()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--no-overlap", action="store_true", help="A/B: the reference-order autograd step as a single-stream graph")
    ap.add_argument("--overlap", action="store_true", help="A/B: the frozen classifier's term on a parallel graph branch (two streams)")
    ap.add_argument("--inline", action="store_true", help="(default) the scheduled step's kernels on ONE stream")
    ap.add_argument("--through-trainer", action="store_true",
                    help="time the iterations THROUGH house.train_countergan(generator, config, X_train, y_train, clf_model) — the reference-shaped "
                         "entry point (17,290 synthetic rows = the reference's training split, 4 iterations per epoch at batch 4096, diagnostics, "
                         "epoch summaries, D_grad variant): marginal seconds per iteration between a short and a long run (setup cancels)")
    ap.add_argument("--eager-draws", action="store_true", help="A/B: the per-step draws as an eager launch in front of each replay (host-side Philox offsets)")
    BL.add_common_args(ap, steps=200, warmup=20)
    args = ap.parse_args(argv)
    if args.gpus != 1:
        sys.exit("bench_house.py: the tabular step does not shard (BASELINE config 5 is single-GPU); run one process per replica")
    R = BL.Ranks(args, os.path.abspath(__file__))
    from pcgan_amd import house as H, ops
    dev = R.dev
    if args.through_trainer:
        return through_trainer(args, R, H, ops, dev)
    G, D, C = H.build(dev, seed=0)
    opt_g, opt_d = H.make_optimizers(G, D)
    norm = H.cat_norm_maps(G, H.CONFIG, dev)
    rng = ops.DeviceRNG(1)
    B = args.batch
    x = rng.rand((B, H.CONFIG["input_dim"]), dev)
    y = rng.randint(0, H.CONFIG["num_classes"], B, dev)
    t, mask, noise = H.draw_batch_randoms(rng, G, y, H.CONFIG, dev)

    onehots = None

    def draws():                      # straight into the step's (static) input buffers, the one-hot rows with them
        H.draw_batch_randoms(rng, G, y, H.CONFIG, dev, out=(t, mask, noise), onehots=onehots)

    gs = None
    if not args.eager:
        gs = H.GraphedTrainStep(G, D, C, opt_g, opt_d, norm, B, overlap=False if args.no_overlap else (True if args.overlap else "inline"),
                                rng=None if args.eager_draws else rng)
        gs.x.copy_(x); gs.y.copy_(y)
        t, mask, noise = gs.target_y, gs.mask, gs.noise
        y = gs.y                       # the graph's own label buffer: the draws kernel reads it for the one-hot rows
        onehots = gs.onehots if gs.branch is not None else None

    def run(i):
        if gs is None or args.eager_draws:      # (default: the draws are the first launch of the replayed graph, offsets from a device counter)
            draws()
        return gs.replay() if gs is not None else H.train_step(G, D, C, opt_g, opt_d, x, y, t, mask, norm, gumbel=noise)

    for i in range(args.warmup):
        run(i)
    dt, out = R.timed(run, args.steps)
    sec = dt / args.steps
    losses = {"D_loss": float(out["D_loss"].item()), "G_loss": float(out["G_loss"].item())}
    if not all(v == v and abs(v) < 1e6 for v in losses.values()):
        sys.exit(f"non-finite losses: {losses}")

    # algorithmic HBM bytes per step
    nG, nD, nC = (sum(p.numel() for p in m.parameters()) for m in (G, D, C))
    hid = int(H.CONFIG.get("hidden_dim", 256)) if isinstance(H.CONFIG, dict) else 256
    nblk = len(getattr(G, "blocks", [])) or 5
    per_row_saved = 4 * (2 * (17 + 4 + 17 + 38) + (2 * nblk + 1) * hid * 2 + 3 * 128)      # inputs/draws, G trunk activations (fwd write + bwd read), D/C hidden rows
    algo_bytes = (4 * (3 * nD + 2 * nG + 2 * nC)          # parameters read: D in 3 passes, G fwd+bwd, frozen C fwd+bwd
                  + 28 * (nG + nD)                        # Adam: p, g read; m, v read+write; p write
                  + B * per_row_saved)
    launches = None
    for tag in ("r04", "r03", "r02"):
        try:
            with open(os.path.join(BL.ROOT, "profiles", f"{tag}_house_launches.json")) as f:
                launches = json.load(f)["launches_per_step"]
            break
        except Exception:
            pass
    ach = algo_bytes / sec / 1e9
    roofline = {"bound": "launch/hbm", "achieved": round(ach, 2), "peak": BL.PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / BL.PEAK_HBM_GBS, 5),
                "traffic": None, "algorithmic_bytes_per_step": int(algo_bytes), "launches_per_step": launches,
                "us_per_launch": None if not launches else round(sec * 1e6 / launches, 2),
                "kernel": "whole step (fused generator segments, fused critic pair, MFMA classifier and weight gradients, Adam x2)"}
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import house_ref as HR                   # the checker's restatement: CPU-baseline leg only
        oG, oD, oC = HR.build(0)
        o_g, o_d = HR.make_optimizers(oG, oD)
        xb, yb, tb, mb, gb = HR.synthetic_batch(B, 0)
        nm = HR.cat_norm_maps()
        med, thr, avail = BL.cpu_median(lambda: HR.house_step(oG, oD, oC, o_g, o_d, xb, yb, tb, mb, gb, nm), steps=15, warmup=3,
                                        threads=args.cpu_threads)
        cpu = {"value": round(B / med, 1), "unit": "rows/sec", "cores": thr, "kind": "port", "cpu_model": BL.cpu_model(),
               "host_cpus_visible": avail,
               "sample": f"median of 15 steps at batch {B} (= the GPU run) after 3 warm-ups; PyTorch-CPU fp32 restatement of "
                         f"house_sales_kc_usa/trainer.py:241-316 (oracle/house_ref.py)"}
    BL.emit({
        "metric": "rows/sec (G+D step) tabular CounteRGAN house_sales_kc_usa; launch/HBM bound",
        "value": round(B / sec, 1), "unit": "rows/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(sec * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"conditional_counteRGAN/house_sales_kc_usa tabular CounteRGAN (FiLM ResidualGenerator, spectral-norm "
                               f"Discriminator, frozen NNClassifier), 17 features, batch {B}, full step incl. per-step device draws",
                   "global_batch": B, "parallelism": "dp1"},
        "roofline": roofline, "cpu_baseline": cpu, "final_losses": losses,
        "launch": "eager" if gs is None else ("hip-graph replay (1 graph"
                                            + (", scheduled step on one stream" if gs.branch == "inline" else
                                               ", classifier term on a parallel branch" if gs.branch is not None else ", reference-order step")
                                            + (") + 1 RNG launch per step drawing into its input buffers" if args.eager_draws else
                                               "; the per-step draws are its first launch, Philox offsets from a device counter)")),
    })
    R.finish()


if __name__ == "__main__":
    main()
