#!/usr/bin/env python3
"""bench.py — images/sec of one full DCGAN G+D training step (mnist_dcgan.py:147-175) on MI355X.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W          (one rank per GPU, RCCL; weak scaling: 512 images per GPU)

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child processes, one per
GPU; the parent never touches the GPU) and relays rank 0's JSON line; under torch.distributed.run it is a rank.

A "step" = D(real) fwd+bwd, G fwd, D(fake.detach()) fwd+bwd, Adam(D), D(fake) fwd, bwd through D into G, Adam(G)
on a synthetic MNIST-shaped batch (U[-1,1) 64x64 images, N(0,1) noise) that is already resident in HBM.  fp32
throughout (v_mfma_f32_32x32x2_f32 for the contractions).  Rank 0 prints ONE JSON line.  Defaults: 50 timed steps after 5
warm-up steps (a 0.55 s timed region: long enough for a utilisation sampler to see the GPU busy).

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
    """(HBM bytes per launch of the implicit-GEMM family, file) from the newest committed rocprofv3 --pmc summary (FETCH_SIZE
    doubled + WRITE_SIZE, collected in separate passes: profiles/rNN_pmc_traffic.json), or (None, None)."""
    for name in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                return round(json.load(f)["_igemm_family"]["hbm_bytes_per_launch"]), "profiles/" + name
        except Exception:
            continue
    return None, None


def count_gpus_sysfs():
    """GPUs of this node WITHOUT touching HIP: KFD topology nodes with simd_count > 0 (CPU nodes have 0).  None if the topology
    is not readable (then the caller has to ask the runtime)."""
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    seen = False
    for d in nodes:
        try:
            with open(os.path.join(base, d, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        seen = True
                        n += int(line.split()[1]) > 0
                        break
        except (OSError, ValueError, IndexError):
            continue
    return n if seen else None


class ClockSampler:
    """Samples the GPU's shader clock (and socket power) from sysfs in a host thread while the timed steps run: hwmon freq1_input
    (Hz) / power1_average|power1_input (uW) of the card that backs this rank's HIP device, else the starred line of pp_dpm_sclk.
    No HIP call, no effect on the stream; every field is None when sysfs exposes nothing."""

    def __init__(self, pci=None, period=0.01):
        """pci = (domain, bus, device) of the HIP device (torch.cuda.get_device_properties: pci_domain_id, pci_bus_id,
        pci_device_id): the DRM card whose sysfs device link ends in that address is sampled; without a match nothing is (a box
        exposes the cards of every GPU of the host, also those this job cannot use)."""
        import glob
        self.period, self.freq, self.power, self._stop, self._th = period, [], [], False, None
        self.dev = None
        if pci is not None:
            want = "%04x:%02x:%02x." % tuple(int(v) for v in pci)
            for c in glob.glob("/sys/class/drm/card[0-9]*/device"):
                try:
                    if os.path.basename(os.path.realpath(c)).lower().startswith(want) and "-" not in os.path.basename(os.path.dirname(c)):
                        self.dev = c
                        break
                except OSError:
                    continue
        self.f_freq = self.f_power = self.f_dpm = None
        if self.dev:
            for h in glob.glob(os.path.join(self.dev, "hwmon/hwmon*")):
                if os.path.exists(os.path.join(h, "freq1_input")):
                    self.f_freq = os.path.join(h, "freq1_input")
                for nm in ("power1_average", "power1_input"):
                    if self.f_power is None and os.path.exists(os.path.join(h, nm)):
                        self.f_power = os.path.join(h, nm)
            if os.path.exists(os.path.join(self.dev, "pp_dpm_sclk")):
                self.f_dpm = os.path.join(self.dev, "pp_dpm_sclk")

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except OSError:
            return None

    def sample(self):
        mhz = None
        if self.f_freq:
            t = self._read(self.f_freq)
            try:
                mhz = float(t) / 1e6 if t else None
            except ValueError:
                mhz = None
        if mhz is None and self.f_dpm:
            for line in (self._read(self.f_dpm) or "").splitlines():
                if line.rstrip().endswith("*"):
                    try:
                        mhz = float(line.split(":")[1].strip().lower().split("mhz")[0])
                    except (IndexError, ValueError):
                        pass
        if mhz:
            self.freq.append(mhz)
        if self.f_power:
            t = self._read(self.f_power)
            try:
                if t:
                    self.power.append(float(t) / 1e6)
            except ValueError:
                pass

    def _loop(self):
        while not self._stop:
            self.sample()
            time.sleep(self.period)

    def start(self):
        if self.f_freq or self.f_dpm or self.f_power:
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()

    def stop(self):
        self._stop = True
        if self._th is not None:
            self._th.join()
        def stats(v):
            if not v:
                return None
            v = sorted(v)
            return {"min": round(v[0], 1), "p50": round(v[len(v) // 2], 1), "max": round(v[-1], 1), "samples": len(v)}
        return {"sclk_mhz": stats(self.freq), "power_w": stats(self.power),
                "source": self.f_freq or self.f_dpm or None, "period_s": self.period}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch=BATCH_PER_GPU, steps=3, threads=None):
    """Reference loop restated on PyTorch-CPU (oracle) at the bench batch: 1 warm-up step, then the median
    of `steps` timed steps — a bounded sample (~10 s of CPU work)."""
    from oracle import dcgan_ref as R
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box gives the job a 16-core share of the host; measured there (r01): 8 thr 221, 16 thr 328, 32 thr 207, 64 thr
    # 87 img/s — more threads than the share make the CPU figure worse, so the default is the share
    cores = threads or min(16, avail)
    torch.set_num_threads(cores)
    netG, netD = R.build(None, seed=1)
    crit, optD, optG = R.make_optimizers(netG, netD)
    real, noise = R.synthetic_batch(batch, seed=0)
    R.dcgan_step(netG, netD, crit, optD, optG, real, noise)
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        R.dcgan_step(netG, netD, crit, optD, optG, real, noise)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[len(times) // 2]
    return {"value": round(batch / med, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model(), "host_cpus_visible": avail,
            "step_seconds": [round(t, 3) for t in times],
            "sample": f"median of {steps} G+D steps at batch {batch} (= the bench batch), after 1 warm-up step; PyTorch-CPU fp32 "
                      f"restatement of mnist_dcgan.py:147-175 (oracle/dcgan_ref.py), {torch.get_num_threads()} threads"}


def secondary_benches(dev, no_calib=False):
    """BASELINE configs 3 / 4 / 5 on the same box, in this process, AFTER the headline's timed region and calibration: a short
    replay of scripts/bench_wgan.py, bench_countergan.py and bench_house.py (their own main(), lines captured instead of printed;
    no CPU baseline).  Each entry carries a bare-MFMA calibration taken right behind it, so `step_over_calib_mfma` does not depend
    on the box's clock.  The headline `value` is not touched by any of this."""
    import importlib
    sdir = os.path.join(ROOT, "scripts")
    if sdir not in sys.path:
        sys.path.insert(0, sdir)
    BL = importlib.import_module("_benchlib")
    from pcgan_amd import ops
    out = {}
    for name, mod, argv in (("wgan", "bench_wgan", ["--steps", "10", "--warmup", "2"]),
                            ("countergan", "bench_countergan", ["--steps", "8", "--warmup", "2"]),
                            ("house", "bench_house", ["--steps", "200", "--warmup", "20"])):
        t0 = time.perf_counter()
        lines = BL.capture(True)
        try:
            importlib.import_module(mod).main(argv + ["--no-cpu-baseline"])
            line = lines[-1] if lines else None
        except SystemExit as e:      # a bench that refuses (non-finite losses, ...) must not take the headline line down with it
            line, err = None, str(e)
        except Exception as e:
            line, err = None, f"{type(e).__name__}: {e}"
        finally:
            BL.capture(False)
            ops.set_conv_hook(None)
        if line is None:
            out[name] = {"error": err if "err" in dir() else "no line"}
            continue
        roof = line.get("roofline") or {}
        ent = {"ms_per_step": line["ms_per_step"], "value": line["value"], "unit": line["unit"], "steps": line["steps"],
               "workload": line["config"]["workload"], "step_frac": roof.get("step_frac"),
               "family_tflops": roof.get("achieved") if roof.get("bound") == "mfma" else None,
               "launches_per_step": roof.get("launches_per_step"), "launch": line.get("launch"),
               "final_losses": line.get("final_losses")}
        for k in ("critic_update_ms", "generator_update_ms"):
            if k in line:
                ent[k] = line[k]
        if not no_calib:
            torch.cuda.synchronize()
            c = ops.calibrate(dev, copy_mb=64)
            ent["calib_mfma_tflops"] = c["mfma_tflops"]
            if ent["step_frac"] is not None:
                ent["step_over_calib_mfma"] = round(ent["step_frac"] * PEAK_F32_MFMA_TFLOPS / c["mfma_tflops"], 4)
            if ent["family_tflops"] is not None:
                ent["family_over_calib_mfma"] = round(ent["family_tflops"] / c["mfma_tflops"], 4)
        ent["seconds_spent"] = round(time.perf_counter() - t0, 1)
        out[name] = ent
    out["what"] = ("BASELINE configs 3 (conditional WGAN-GP, width 1024, batch 256), 4 (CounteRGAN/mnist, batch 1024) and 5 (house-sales "
                   "tabular CounteRGAN, batch 4096) timed on this box after the headline: scripts/bench_{wgan,countergan,house}.py run "
                   "in-process with the step counts above; same contract (inputs resident, graph replay, barrier + synchronize brackets)")
    return out


_json_fd = None


def claim_stdout():
    """Keep the process's stdout for the ONE JSON line: the real fd 1 is saved and fd 1 is pointed at stderr, so that anything a
    library prints there (RCCL writes a version banner to stdout when a communicator is created) cannot land in front of it."""
    global _json_fd
    if _json_fd is None:
        sys.stdout.flush()
        _json_fd = os.dup(1)
        os.dup2(2, 1)


def emit_json(line):
    text = (json.dumps(line) + "\n").encode()
    if _json_fd is None:
        sys.stdout.write(text.decode()); sys.stdout.flush()
    else:
        os.write(_json_fd, text)


def launch_ranks(n, argv, script=None):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as CHILD processes — one per GPU, the
    environment torch.distributed.run would give them — before anything in this process touches the GPU, relay rank 0's stdout
    (the JSON line) and fail if any rank fails.  The parent counts GPUs from the KFD topology in sysfs (no HIP initialisation),
    prefixes every rank's stderr with its rank, and gives up (non-zero) if the ranks have not all exited `PCG_BENCH_TIMEOUT`
    seconds (default 900) after the start — a rank stuck in the rendezvous cannot hang the run."""
    import socket
    import subprocess
    have = count_gpus_sysfs()                 # KFD topology: no HIP initialisation in the parent
    if have is None:
        have = torch.cuda.device_count()      # topology unreadable: ask the runtime (the children are fresh processes either way)
    if have < n:
        sys.exit(f"bench.py --gpus {n}: this node exposes {have} GPU(s); one rank per GPU is required (no oversubscription)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import threading
    limit = float(os.environ.get("PCG_BENCH_TIMEOUT", "900"))

    def relay(r, pipe):
        # a rank's stderr (and, for ranks > 0, stdout) line by line with the rank in front: interleaved tracebacks stay readable
        for raw in iter(pipe.readline, b""):
            sys.stderr.write(f"[rank {r}] " + raw.decode("utf-8", "replace"))
            sys.stderr.flush()
        pipe.close()

    procs, relays = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # rank 0 inherits stdout (its JSON line is the bench output); the other ranks print nothing there, their stdout joins their stderr
        if r == 0:
            p = subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env, stderr=subprocess.PIPE)
        else:
            p = subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        t = threading.Thread(target=relay, args=(r, p.stderr if r == 0 else p.stdout), daemon=True)
        t.start()
        procs.append(p); relays.append(t)
    rc = 0
    t_start = time.monotonic()
    pending = dict(enumerate(procs))
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in pending.values():      # exactly the processes started above
                    q.terminate()
        if pending and time.monotonic() - t_start > limit:
            print(f"[bench] ranks {sorted(pending)} still running after {limit:.0f} s (PCG_BENCH_TIMEOUT): a rank that never reached "
                  f"the rendezvous, or a hung collective; stopping them", file=sys.stderr, flush=True)
            for q in pending.values():
                q.terminate()
            deadline = time.monotonic() + 10
            for q in pending.values():
                try:
                    q.wait(max(0.1, deadline - time.monotonic()))
                except subprocess.TimeoutExpired:
                    q.kill()
            pending.clear()
            rc = rc or 124
        time.sleep(0.05)
    for t in relays:
        t.join(5)
    sys.exit(rc)


def _read_tail(path, limit=2000):
    """What RCCL wrote for this rank under NCCL_DEBUG=WARN (empty = no warning), or None when the log was redirected elsewhere."""
    if not path:
        return None
    try:
        with open(path, errors="replace") as f:
            return f.read()[-limit:]
    except OSError:
        return ""


def params_digest(nets):
    """One int64 per net: the wrap-around sum of the parameter bits — equal on two replicas iff (up to a 2^-64 collision) their
    flat parameter buffers are bit-identical.  Runs after the timed region (ATen reduction: diagnostics, not the step)."""
    return torch.stack([n.flat_params.view(torch.int32).to(torch.int64).sum() for n in nets])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="images per GPU (default: the BASELINE config, 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket conv launches with HIP events")
    ap.add_argument("--force-dp", action="store_true", help="exercise the RCCL gradient-sync path even with one rank")
    ap.add_argument("--eager", action="store_true", help="enqueue every step kernel by kernel instead of replaying the captured HIP graph(s)")
    ap.add_argument("--event-every", type=int, default=8, help="one timed step per this many (at most 3 in total) runs eagerly with HIP events around the conv launches")
    ap.add_argument("--sync-bn", action="store_true", help="exact global-batch BatchNorm across the ranks (SURVEY.md 8e option ii); implies --eager")
    ap.add_argument("--wgrad-stream", action="store_true", help="A/B: weight gradients on a second HIP stream beside the grad-input kernels")
    ap.add_argument("--torch-collectives", action="store_true", help="A/B: gradient exchange on torch.distributed's RCCL communicator instead of the library's (pcg_dp_*)")
    ap.add_argument("--no-calib", action="store_true", help="skip the bare-MFMA / HBM-copy calibration around the timed region")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short replays of BASELINE configs 3 / 4 / 5 behind the headline (the `secondary` block)")
    ap.add_argument("--cpu-threads", type=int, default=None, help="threads of the CPU baseline (default: min(16, visible cores))")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args.gpus, sys.argv[1:])      # does not return
    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world} (one rank per GPU; start it with "
                 f"--nproc-per-node {args.gpus} or let `python bench.py --gpus N` start the ranks itself)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import pcgan_amd
    from pcgan_amd import dcgan as D, ops
    pcgan_amd.load()

    dp = None
    rccl_ranks = None
    rccl_log = None
    if world > 1 or args.force_dp:
        import tempfile
        if os.environ.get("NCCL_DEBUG", "VERSION").upper() == "VERSION":  # RCCL's warnings of every rank (first contact with N > 1);
            os.environ["NCCL_DEBUG"] = "WARN"                             # an explicit INFO / TRACE from the caller is left alone
        if "NCCL_DEBUG_FILE" not in os.environ:
            rccl_log = os.path.join(tempfile.gettempdir(), f"pcg_bench_rccl_{os.getpid()}_rank{rank}.log")
            os.environ["NCCL_DEBUG_FILE"] = rccl_log
        import torch.distributed as dist
        from pcgan_amd.parallel import GradSync, broadcast_parameters
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dp = GradSync(always_exchange=args.force_dp, native=not args.torch_collectives, sync_bn=args.sync_bn)
        if args.sync_bn:
            args.eager = True
        rccl_ranks = dp.rccl_ranks()                # ranks of the communicator that carries the gradient exchange (the driver checks it against --gpus)

    # random-init weights of the reference architecture (weights_init distribution), identical on every rank
    torch.manual_seed(1)
    netG, netD = D.build(None, device="cpu")
    netG.to(dev); netD.to(dev)
    crit, optD, optG = D.make_optimizers(netG, netD)
    if args.wgrad_stream:
        netG.wgrad_stream = netD.wgrad_stream = torch.cuda.Stream()
    if dp is not None:
        netG.flat_params, netD.flat_params  # flatten
        broadcast_parameters(netG, dp=dp); broadcast_parameters(netD, dp=dp)

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

    def barrier():          # on the communicator that carries the exchange (pcg_dp_barrier): the timed region uses ONE communicator
        if dp is not None:
            dp.barrier()

    # Calibration, OUTSIDE the timed bracket (before the warm-up steps and after the timed ones): a ~20 ms bare fp32-MFMA loop
    # on every CU and a 512 MiB device copy — what this box's matrix pipe and HBM sustain now, so that a slower box can be told
    # from a slower kernel (roofline.achieved / calib.mfma_tflops does not depend on the box's clock).
    calib = None
    if not args.no_calib:
        calib = {"before": ops.calibrate(dev)}
        torch.cuda.synchronize()

    if args.warmup > 0 and not args.eager and not args.no_kernel_events:
        out = step(0, eager=True)      # untimed: fills this stream's allocator pool for the event-sampled eager steps of the timed region
    for i in range(args.warmup):
        out = step(i)
    if dp is not None:
        dp.wait_all()
    torch.cuda.synchronize()

    # HIP events around every launch of the implicit-GEMM family, on a sample of the timed steps (those run eagerly; events
    # cannot be read back from inside a graph): bracketing all ~35 launches costs ~3 % of a step, sampling keeps it < 1 %
    records = []
    hook = (lambda label, flops, t0, t1: records.append((label, flops, t0, t1))) if not args.no_kernel_events else None

    sampler = None
    if rank == 0:
        pr = torch.cuda.get_device_properties(local_rank)
        sampler = ClockSampler((getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1), getattr(pr, "pci_device_id", 0)))
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]     # one record per step boundary: the spread
    barrier()
    torch.cuda.synchronize()
    if sampler is not None:
        sampler.start()
    t0 = time.perf_counter()
    marks[0].record()
    # steps that run eagerly with events: one per --event-every timed steps, at most 3, evenly spread (>= 1: short runs still carry a
    # live roofline measurement); 2 of the default 20
    nsamp = 0 if hook is None or args.steps <= 0 else max(1, min(3, args.steps // max(1, args.event_every)))
    sampled = sorted({min(args.steps - 1, ((2 * j + 1) * args.steps) // (2 * nsamp)) for j in range(nsamp)})
    for i in range(args.steps):
        timed = i in sampled
        ops.set_conv_hook(hook if timed else None)
        out = step(args.warmup + i, eager=timed)
        marks[i + 1].record()
    host_enqueue = time.perf_counter() - t0      # the host is done issuing; the GPU is still working if this < elapsed
    if dp is not None:
        dp.wait_all()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.set_conv_hook(None)
    clocks = sampler.stop() if sampler is not None else None
    if calib is not None:
        calib["after"] = ops.calibrate(dev)
        torch.cuda.synchronize()
    # per-step spread from the boundary events (replayed and event-sampled eager steps told apart: the eager ones carry ~35 event
    # pairs and the host's launch latency); `value` stays the mean over the whole bracket, as the contract defines it
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]

    if dp is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # data-parallel replicas must hold bit-identical parameters after the timed steps (same start, same averaged gradients,
    # same Adam): all-gather a digest of both nets' flat parameter buffers and compare
    replicas_identical = None
    if dp is not None:
        dig = params_digest([netG, netD])
        allg = [torch.empty_like(dig) for _ in range(world)]
        torch.distributed.all_gather(allg, dig)
        replicas_identical = all(torch.equal(a, allg[0]) for a in allg)
        if not replicas_identical:
            sys.exit(f"rank {rank}: replicas diverged — parameter digests {[a.tolist() for a in allg]}")

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
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": pmc_traffic()[0],
            "traffic_unit": f"HBM bytes per launch (rocprofv3 PMC, {pmc_traffic()[1]})",
            "kernel": "fp32-MFMA implicit-GEMM conv family (igemm_mainloop: conv_fwd/dgrad/wgrad_kernel); wgrad spans include slab_reduce",
            "step_frac": round(ALGO_GFLOP_PER_IMAGE * 1e9 * args.batch / (elapsed / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "gemm_time_share": round(tot_t / (elapsed * len(sampled) / args.steps), 4),
            "per_kernel": {k: {"launches": a[0], "avg_ms": round(a[2] / a[0] * 1e3, 4),
                               "tflops": round(a[1] / a[2] / 1e12, 2)} for k, a in sorted(agg.items())},
        }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(batch=args.batch, threads=args.cpu_threads)

    second = None
    if rank == 0 and world == 1 and dp is None and not args.no_secondary:
        second = secondary_benches(dev, no_calib=args.no_calib)

    spread = None
    if step_ms:
        def _stats(v):
            v = sorted(v)
            return {"min": round(v[0], 3), "p50": round(v[len(v) // 2], 3), "max": round(v[-1], 3), "n": len(v)}
        replayed = [t for i, t in enumerate(step_ms) if i not in sampled]
        spread = {"all": _stats(step_ms), "replayed": _stats(replayed) if replayed else None,
                  "event_sampled_eager": _stats([step_ms[i] for i in sampled]) if sampled else None}
    if calib is not None:
        b, a = calib["before"], calib["after"]
        calib["mfma_tflops_before"], calib["mfma_tflops_after"] = b["mfma_tflops"], a["mfma_tflops"]
        calib["mfma_tflops"] = round(0.5 * (b["mfma_tflops"] + a["mfma_tflops"]), 2)
        calib["hbm_gbs"] = round(0.5 * (b["hbm_gbs"] + a["hbm_gbs"]), 1)
        calib["mfma_clock_mhz"] = round(0.5 * (b["mfma_clock_mhz"] + a["mfma_clock_mhz"]))
        calib["what"] = ("pcg_calib_mfma: bare v_mfma_f32_32x32x2_f32 loop on every CU (~20 ms), clock from in-kernel s_memtime / "
                         "s_memrealtime, behind ~30 ms of the same loop untimed (no cold-clock reading); pcg_calib_copy: 512 MiB device "
                         "copy (read + write bytes); run before the warm-up and after the timed steps, outside the timed bracket")
        if roofline is not None:
            roofline["achieved_over_calib_mfma"] = round(roofline["achieved"] / calib["mfma_tflops"], 4)
            roofline["step_over_calib_mfma"] = round(ALGO_GFLOP_PER_IMAGE * 1e9 * args.batch / (elapsed / args.steps) / 1e12
                                                     / calib["mfma_tflops"], 4)

    if rank == 0:
        line = {
            "metric": "images/sec (G+D step) MNIST-28 cDCGAN bs512 @1/2/4/8 MI355X; % MFMA roofline",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "dconv_gan/mnist DCGAN 64x64 (MNIST 28->64), z=100, g_hidden=d_hidden=64, "
                                   f"batch {args.batch} per GPU, full G+D step incl. BatchNorm, BCE, Adam x2",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}"},
            "roofline": roofline, "cpu_baseline": cpu, "calib": calib, "step_ms": spread, "clocks_during_timed_region": clocks,
            "final_losses": losses, "secondary": second,
            "rccl_ranks": rccl_ranks, "replicas_identical": replicas_identical,
            "batchnorm": None if dp is None else ("global batch (statistic sums all-reduced)" if dp.sync_bn else "per replica"),
            "collectives": None if dp is None else {
                "backend": "libpcgan_hip pcg_dp_* (RCCL behind the C ABI)" if dp.native else "torch.distributed nccl",
                "rccl_version": dp.rccl_version(), "timed_region_barrier": "pcg_dp_barrier (same communicator)" if dp.native else "torch.distributed.barrier",
                "nccl_debug": os.environ.get("NCCL_DEBUG"), "rank0_rccl_log": _read_tail(rccl_log)},
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
            "launch": "eager" if gs is None else f"hip-graph replay ({len(gs.program)} segment(s)); {len(sampled)} of {args.steps} timed steps eager with HIP events",
        }
        emit_json(line)
    if dp is not None:
        from pcgan_amd.parallel import shutdown
        torch.cuda.synchronize()
        shutdown()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
