"""Shared plumbing of the secondary benches (scripts/bench_countergan.py, bench_wgan.py, bench_house.py): the same contract as
bench.py — `--gpus N` starts its own ranks (bench.launch_ranks) or runs as a rank under torch.distributed.run, a barrier +
synchronize on both sides of the timed region, MAX over ranks, ONE JSON line from rank 0 with `roofline` and `cpu_baseline`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

import bench as _contract  # noqa: E402  (launch_ranks, cpu_model, params_digest)

PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def add_common_args(ap, steps, warmup):
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=steps)
    ap.add_argument("--warmup", type=int, default=warmup)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dp", action="store_true", help="exercise the RCCL gradient-sync path even with one rank")
    ap.add_argument("--eager", action="store_true", help="no HIP-graph replay")
    ap.add_argument("--cpu-threads", type=int, default=None)


class Ranks:
    """Rank bootstrap: returns after torch.distributed + GradSync are up (or with dp None for a plain single-GPU run)."""

    def __init__(self, args, script):
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            _contract.launch_ranks(args.gpus, sys.argv[1:], script=script)      # does not return
        if _capture is None:
            _contract.claim_stdout()
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            sys.exit("needs an MI355X: torch.cuda.is_available() is False (there is no CPU path)")
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        import pcgan_amd
        pcgan_amd.load()
        self.dp = None
        if self.world > 1 or args.force_dp:
            import torch.distributed as dist
            from pcgan_amd.parallel import GradSync
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29513")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            self.dp = GradSync(always_exchange=args.force_dp)

    def broadcast(self, nets):
        if self.dp is not None:
            from pcgan_amd.parallel import broadcast_parameters
            for n in nets:
                n.flat_params      # flatten
                broadcast_parameters(n, dp=self.dp)

    def barrier(self):
        if self.dp is not None:
            self.dp.barrier()

    def timed(self, fn, steps):
        """barrier + sync, `steps` calls of fn(i), drain, barrier; returns (seconds: MAX over ranks, last result)."""
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            out = fn(i)
        if self.dp is not None:
            self.dp.wait_all()
        torch.cuda.synchronize()
        self.barrier()
        dt = time.perf_counter() - t0
        if self.dp is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, out

    def replicas_identical(self, nets):
        if self.dp is None:
            return None
        dig = _contract.params_digest(nets)
        allg = [torch.empty_like(dig) for _ in range(self.world)]
        torch.distributed.all_gather(allg, dig)
        same = all(torch.equal(a, allg[0]) for a in allg)
        if not same:
            sys.exit(f"rank {self.rank}: replicas diverged — parameter digests {[a.tolist() for a in allg]}")
        return same

    def finish(self):
        if self.dp is not None:
            from pcgan_amd.parallel import shutdown
            torch.cuda.synchronize()
            shutdown()
            torch.distributed.destroy_process_group()


def pmc_traffic(name):
    """(HBM bytes per launch of the implicit-GEMM family, file) from the newest committed rocprofv3 --pmc summary of this bench
    (FETCH_SIZE doubled + WRITE_SIZE, separate passes: profiles/r0N_pmc_traffic_<name>.json), or (None, None)."""
    import json
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    for tag in ("r04", "r03", "r02"):
        try:
            fn = f"{tag}_pmc_traffic_{name}.json"
            with open(os.path.join(root, fn)) as f:
                return round(json.load(f)["_igemm_family"]["hbm_bytes_per_launch"]), fn
        except Exception:
            continue
    return None, None


def conv_family_roofline(records, step_seconds, sampled_steps, algo_flops_per_step, traffic=None, traffic_file=None):
    """`records`: (label, flops, start_event, end_event) of every MFMA conv launch in the event-sampled eager steps (ops.set_conv_hook)."""
    agg = {}
    for label, flops, e0, e1 in records:
        if label.startswith("thin"):
            continue
        a = agg.setdefault(label, [0, 0.0, 0.0])
        a[0] += 1; a[1] += flops; a[2] += e0.elapsed_time(e1) * 1e-3
    tot_f = sum(a[1] for a in agg.values()); tot_t = sum(a[2] for a in agg.values())
    if tot_t <= 0:
        return None
    ach = tot_f / tot_t / 1e12
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": f"HBM bytes per launch (rocprofv3 PMC, profiles/{traffic_file})" if traffic_file else None,
            "kernel": "fp32-MFMA implicit-GEMM conv family (conv_fwd / dgrad / wgrad kernels; HIP events around every launch)",
            "step_frac": round(algo_flops_per_step / step_seconds / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "gemm_time_share": round(tot_t / (step_seconds * sampled_steps), 4),
            "per_kernel": {k: {"launches": a[0], "avg_ms": round(a[2] / a[0] * 1e3, 4), "tflops": round(a[1] / a[2] / 1e12, 2)}
                           for k, a in sorted(agg.items())}}


def cpu_median(step_fn, steps=3, warmup=1, threads=None):
    """(median seconds per call, threads) of a CPU oracle step: bounded sample, 1 warm-up."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(threads or min(16, avail))
    for _ in range(warmup):
        step_fn()
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        step_fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], torch.get_num_threads(), avail


_capture = None      # a list: emit() appends the line there instead of printing it (bench.py's `secondary` block runs the benches in-process)


def capture(on=True):
    """Start (returns the list the lines go to) or stop capturing emit()."""
    global _capture
    _capture = [] if on else None
    return _capture


def emit(line):
    if _capture is not None:
        _capture.append(line)
    else:
        _contract.emit_json(line)


cpu_model = _contract.cpu_model
