#!/usr/bin/env python3
"""bench.py — images/sec of one full DCGAN G+D training step (mnist_dcgan.py:147-175) on MI355X.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (one rank per GPU, RCCL; weak scaling: 512 images per GPU)

A "step" = D(real) fwd+bwd, G fwd, D(fake.detach()) fwd+bwd, Adam(D), D(fake) fwd, bwd through D into G, Adam(G)
on a synthetic MNIST-shaped batch (U[-1,1) 64x64 images, N(0,1) noise) that is already resident in HBM.  fp32
throughout (v_mfma_f32_32x32x2_f32 for the contractions).  Rank 0 prints ONE JSON line.

roofline: the dominant kernels are the fp32-MFMA implicit-GEMM convolutions (conv_{fwd,dgrad,wgrad}_kernel).  Every
launch of that family inside the timed region is bracketed by HIP events on the launch stream; `achieved` =
sum of algorithmic FLOPs (2*B*OH*OW*Cout*KH*KW*Cin per launch) / sum of event-measured durations (the timed steps that run
eagerly: one per --event-every steps, at most three; the others replay the captured HIP graph of the same step); `peak` = 157.3
TFLOP/s (fp32 MFMA, MI355X_MICROARCH.md).  `step_frac` = algorithmic FLOPs of the whole step (2.237 GFLOP/image,
SURVEY.md §8d) / wall time / peak — the number the 50 % target is stated on.
cpu_baseline: the oracle restatement of the reference loop (oracle/dcgan_ref.py, PyTorch CPU fp32) timed on the host
cores of this box, rank 0, N=1 only, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

BATCH_PER_GPU = 512
ALGO_GFLOP_PER_IMAGE = 2.236874752      # 2 * 1,118,437,376 MACs (required set)
PEAK_F32_MFMA_TFLOPS = 157.3


def pmc_traffic():
    """HBM bytes per launch of the implicit-GEMM family from the committed rocprofv3 --pmc summary (FETCH_SIZE doubled
    + WRITE_SIZE, collected in separate passes: profiles/r01_pmc_traffic.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return round(json.load(f)["_igemm_family"]["hbm_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(batch=256, steps=2):
    """Reference loop restated on PyTorch-CPU (oracle), bounded sample: 1 warm-up + `steps` timed steps."""
    from oracle import dcgan_ref as R
    # the GPU box gives one GPU a 16-core share; measured there: 8 thr 221, 16 thr 328, 32 thr 207, 64 thr 87 img/s
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    netG, netD = R.build(None, seed=1)
    crit, optD, optG = R.make_optimizers(netG, netD)
    real, noise = R.synthetic_batch(batch, seed=0)
    R.dcgan_step(netG, netD, crit, optD, optG, real, noise)
    t0 = time.perf_counter()
    for _ in range(steps):
        R.dcgan_step(netG, netD, crit, optD, optG, real, noise)
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} G+D steps at batch {batch} (bench batch {BATCH_PER_GPU}), 1 warm-up step, PyTorch-CPU fp32 "
                      f"restatement of mnist_dcgan.py:147-175 (oracle/dcgan_ref.py)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="images per GPU (default: the BASELINE config, 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--force-dp", action="store_true", help="exercise the RCCL gradient-sync path even with one rank")
    ap.add_argument("--eager", action="store_true", help="enqueue every step kernel by kernel instead of replaying the captured HIP graph(s)")
    ap.add_argument("--event-every", type=int, default=8, help="one timed step per this many (at most 3 in total) runs eagerly with HIP events around the conv launches")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N>1 must be launched by torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import pcgan_amd
    from pcgan_amd import dcgan as D, ops
    pcgan_amd.load()

    dp = None
    if world > 1 or args.force_dp:
        import torch.distributed as dist
        from pcgan_amd.parallel import GradSync, broadcast_parameters
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dp = GradSync(always_exchange=args.force_dp)

    # random-init weights of the reference architecture (weights_init distribution), identical on every rank
    torch.manual_seed(1)
    netG, netD = D.build(None, device="cpu")
    netG.to(dev); netD.to(dev)
    crit, optD, optG = D.make_optimizers(netG, netD)
    if dp is not None:
        netG.flat_params, netD.flat_params  # flatten
        broadcast_parameters(netG); broadcast_parameters(netD)

    # synthetic MNIST-shaped batches, resident in HBM before the timed region; each rank its own shard
    g = torch.Generator().manual_seed(1234 + rank)
    nbatches = 4
    reals = [(torch.rand(args.batch, 1, 64, 64, generator=g) * 2 - 1).to(dev) for _ in range(nbatches)]
    noises = [torch.randn(args.batch, 100, 1, 1, generator=g).to(dev) for _ in range(nbatches)]

    def eager_step(i):
        return D.train_step(netG, netD, crit, optD, optG, reals[i % nbatches], noises[i % nbatches], dp=dp)

    # The step is captured once as HIP graph(s) and replayed (segments cut at the gradient exchanges when data-parallel): the
    # kernels and their order are those of eager_step, the host issues 1-4 calls per step instead of ~130 launches, so a slow or
    # shared host cannot starve the GPU.  The next batch is copied into the graph's static input buffers inside the timed region.
    gs = None
    if not args.eager:
        from pcgan_amd.nn import GraphedStep
        s_real, s_noise = reals[0].clone(), noises[0].clone()
        try:
            if dp is None:
                gs = GraphedStep(lambda: D.train_step(netG, netD, crit, optD, optG, s_real, s_noise), {"real": s_real, "noise": s_noise},
                                 [netG, netD], [optD, optG])
            else:
                gs = GraphedStep(lambda d: D.train_step(netG, netD, crit, optD, optG, s_real, s_noise, dp=d),
                                 {"real": s_real, "noise": s_noise}, [netG, netD], [optD, optG], dp=dp)
        except Exception as e:   # capture is an optimisation of the host side only: the same step runs eagerly (every rank takes the
            gs = None            # same path: the failure modes are deterministic — an unsupported capture of some runtime call)
            print(f"[bench] rank {rank}: HIP-graph capture failed ({type(e).__name__}: {e}); running eager launches", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            if dp is not None:
                dp.wait_all()

    def step(i, eager=False):
        if gs is None or eager:
            return eager_step(i)
        gs.load(real=reals[i % nbatches], noise=noises[i % nbatches])
        return gs.replay()

    def barrier():
        if dp is not None:
            torch.distributed.barrier()

    for i in range(args.warmup):
        out = step(i)
    if dp is not None:
        dp.wait_all()
    torch.cuda.synchronize()

    # HIP events around every launch of the implicit-GEMM family, on a sample of the timed steps (those run eagerly; events
    # cannot be read back from inside a graph): bracketing all ~35 launches costs ~3 % of a step, sampling keeps it < 1 %
    records = []
    hook = (lambda label, flops, t0, t1: records.append((label, flops, t0, t1))) if not args.no_kernel_events else None

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # steps that run eagerly with events: one per --event-every timed steps, at most 3, evenly spread (>= 1: short runs still carry a
    # live roofline measurement); 2 of the default 20
    nsamp = 0 if hook is None or args.steps <= 0 else max(1, min(3, args.steps // max(1, args.event_every)))
    sampled = sorted({min(args.steps - 1, ((2 * j + 1) * args.steps) // (2 * nsamp)) for j in range(nsamp)})
    for i in range(args.steps):
        timed = i in sampled
        ops.set_conv_hook(hook if timed else None)
        out = step(args.warmup + i, eager=timed)
    host_enqueue = time.perf_counter() - t0      # the host is done issuing; the GPU is still working if this < elapsed
    if dp is not None:
        dp.wait_all()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.set_conv_hook(None)

    if dp is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    losses = {k: float(out[k].item()) for k in ("errD_real", "errD_fake", "errG")}
    if not all(v == v and abs(v) < 1e4 for v in losses.values()):
        sys.exit(f"non-finite losses after the timed region: {losses}")

    images = world * args.batch * args.steps
    value = images / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    roofline = None
    if records:
        agg = {}
        for label, flops, e0, e1 in records:
            if label.startswith("thin"):
                continue
            a = agg.setdefault(label, [0, 0.0, 0.0])
            a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) * 1e-3
        tot_f = sum(a[1] for a in agg.values()); tot_t = sum(a[2] for a in agg.values())
        achieved = tot_f / tot_t / 1e12
        roofline = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": pmc_traffic(),
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/r01_pmc_traffic.json)",
            "kernel": "fp32-MFMA implicit-GEMM conv family (igemm_mainloop: conv_fwd/dgrad/wgrad_kernel); wgrad spans include slab_reduce",
            "step_frac": round(ALGO_GFLOP_PER_IMAGE * 1e9 * args.batch / (elapsed / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "gemm_time_share": round(tot_t / (elapsed * len(sampled) / args.steps), 4),
            "per_kernel": {k: {"launches": a[0], "avg_ms": round(a[2] / a[0] * 1e3, 4),
                               "tflops": round(a[1] / a[2] / 1e12, 2)} for k, a in sorted(agg.items())},
        }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        line = {
            "metric": "images/sec (G+D step) MNIST-28 cDCGAN bs512 @1/2/4/8 MI355X; % MFMA roofline",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "dconv_gan/mnist DCGAN 64x64 (MNIST 28->64), z=100, g_hidden=d_hidden=64, "
                                   f"batch {args.batch} per GPU, full G+D step incl. BatchNorm, BCE, Adam x2",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "final_losses": losses,
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
            "launch": "eager" if gs is None else f"hip-graph replay ({len(gs.program)} segment(s)); {len(sampled)} of {args.steps} timed steps eager with HIP events",
        }
        print(json.dumps(line), flush=True)
    if dp is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
