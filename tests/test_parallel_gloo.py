"""CPU, world_size 2 over gloo: the data-parallel gradient exchange (one flat bucket per net, averaged across ranks,
replicas stay identical).  The GPU path uses the same GradSync object with backend nccl (= RCCL)."""
import os
import time
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeNet:
    def __init__(self, n, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.flat_grads = torch.randn(n, generator=g)
        self.flat_params = torch.full((n,), float(rank))
        self._bufs = [torch.full((3,), float(rank))]

    def buffers(self):
        return self._bufs


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcgan_amd.parallel import GradSync, broadcast_parameters
    netD, netG = _FakeNet(1001, rank), _FakeNet(77, rank)
    expD = sum(_FakeNet(1001, r).flat_grads for r in range(world)) / world
    expG = sum(_FakeNet(77, r).flat_grads for r in range(world)) / world
    dp = GradSync()
    dp.sync_now(netD)
    stepped = []
    dp.sync_then(netG, lambda: stepped.append(netG.flat_grads.clone()))
    dp.wait(netG)
    dp.wait_all()
    broadcast_parameters(netD, src=0)
    ok = (torch.allclose(netD.flat_grads, expD, atol=1e-6) and torch.allclose(netG.flat_grads, expG, atol=1e-6)
          and len(stepped) == 1 and torch.equal(stepped[0], netG.flat_grads)
          and float(netD.flat_params.abs().sum()) == 0.0 and float(netD.buffers()[0].abs().sum()) == 0.0)
    # replicas are bit-identical after the exchange
    gathered = [torch.empty_like(netD.flat_grads) for _ in range(world)]
    dist.all_gather(gathered, netD.flat_grads)
    ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)], res


# ---- the data-parallel SEGMENT PROGRAM (nn.GraphedStep + nn._CutDP) on two gloo ranks --------------------------------------
# On the GPU the step is captured as HIP-graph segments cut at the gradient exchanges.  Here the same GraphedStep / _CutDP code
# runs with a recording capture backend: "kernels" are thunks that are recorded (not executed) while a segment is open and run
# on replay — which is exactly the contract of stream capture — and the nets are flat CPU buffers.
class _Recorder:
    def __init__(self):
        self.open = None
        self.captured_segments = 0

    def launch(self, fn):
        if self.open is not None:
            self.open.append(fn)
        else:
            fn()

    # capture-backend interface of nn.GraphedStep
    def warmup(self, run, n, dp):
        for _ in range(n):
            run()
        if dp is not None:
            dp.wait_all()

    def begin(self):
        assert self.open is None
        self.open = []

    def end(self):
        seg, self.open = self.open, None
        self.captured_segments += 1

        class _Seg:
            def replay(_self):
                for fn in seg:
                    fn()
        return _Seg()

    abort = end


class _ToyNet:
    """Flat parameter / gradient buffers with the FlatModule surface GraphedStep and GradSync use."""

    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.flat_params = torch.randint(-8, 8, (n,), generator=g).float()
        self.flat_grads = torch.zeros(n)
        self.bn_running = torch.zeros(4)

    def _ensure_flat(self):
        pass

    def buffers(self):
        return [self.bn_running]


class _ToySGD:
    def __init__(self, net, rec):
        self.net, self.rec, self.steps = net, rec, torch.zeros(1)

    def step(self):
        def k():
            self.net.flat_params.sub_(0.25 * self.net.flat_grads)
            self.steps.add_(1)
        self.rec.launch(k)

    def snapshot(self):
        return self.steps.clone()

    def restore(self, s):
        self.steps.copy_(s)


def _toy_step(rec, netG, netD, optD, optG, real, noise, dp):
    """Same statement order and the same three exchange points as dcgan.train_step (mnist_dcgan.py:147-175)."""
    L = rec.launch
    L(lambda: netD.flat_grads.zero_())                                        # netD.zero_grad()
    L(lambda: netD.flat_grads.add_(real.sum() * netD.flat_params))           # D(real) forward + backward
    if dp is not None:
        dp.wait(netG)                                                         # previous Adam(G) done
    L(lambda: netD.flat_grads.add_(noise.sum() * netG.flat_params[:netD.flat_params.numel()]))   # G(z), D(fake.detach()) fwd+bwd
    L(lambda: netD.bn_running.add_(real.mean()))                              # BatchNorm running statistics: per replica
    if dp is not None:
        dp.sync_now(netD)
    optD.step()
    L(lambda: netG.flat_grads.zero_())                                        # netG.zero_grad()
    L(lambda: netG.flat_grads.add_(noise.sum() * netG.flat_params).add_(netD.flat_params.sum()))   # D(fake) fwd, bwd into G
    if dp is not None:
        dp.sync_then(netG, optG.step)
    else:
        optG.step()
    return netD.flat_grads


def _toy_pair_step(rec, netG, netD, optD, optG, real, noise, dp):
    """The exchange points of dcgan.train_step(pair=True) (r04): wait(G) leads the step — G's forward comes first and the real and fake
    discriminator passes are one pass — then sync_now(D), Adam(D), the G step, sync_then(G, Adam).  Same arithmetic as _toy_step."""
    L = rec.launch
    if dp is not None:
        dp.wait(netG)                                                         # previous Adam(G) done: G's forward leads
    L(lambda: netD.flat_grads.zero_())
    L(lambda: netD.flat_grads.add_(real.sum() * netD.flat_params).add_(noise.sum() * netG.flat_params[:netD.flat_params.numel()]))   # ONE 2B pass
    L(lambda: netD.bn_running.add_(real.mean()))
    if dp is not None:
        dp.sync_now(netD)
    optD.step()
    L(lambda: netG.flat_grads.zero_())
    L(lambda: netG.flat_grads.add_(noise.sum() * netG.flat_params).add_(netD.flat_params.sum()))
    if dp is not None:
        dp.sync_then(netG, optG.step)
    else:
        optG.step()
    return netD.flat_grads


def _toy_countergan_step(rec, netG, netD, optD, optG, real, noise, dp):
    """The exchange points of countergan.train_step (mnist/trainer.py:96-123) in its data-parallel order: D(real) forward hoisted
    above wait(G), generator forward, D update (bucket averaged in stream order), G update (bucket + Adam overlapped)."""
    L = rec.launch
    scratch = {}
    if dp is not None:
        L(lambda: scratch.__setitem__("d_real", real.sum() * netD.flat_params))       # D(real) forward: reads only D
        dp.wait(netG)
    else:
        L(lambda: scratch.__setitem__("d_real", real.sum() * netD.flat_params))
    L(lambda: scratch.__setitem__("x_cf", noise.sum() * netG.flat_params[:netD.flat_params.numel()]))   # generator forward
    L(lambda: netD.flat_grads.zero_())
    L(lambda: netD.flat_grads.add_(scratch["d_real"]).add_(scratch["x_cf"]))          # d_loss.backward()
    if dp is not None:
        dp.sync_now(netD)
    optD.step()
    L(lambda: netG.flat_grads.zero_())
    L(lambda: netG.flat_grads.add_(noise.sum() * netG.flat_params).add_(netD.flat_params.sum()))      # g_loss.backward()
    if dp is not None:
        dp.sync_then(netG, optG.step)
    else:
        optG.step()
    return netD.flat_grads


def _segment_worker(rank, world, port, q, toy="dcgan"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcgan_amd.nn import GraphedStep
    from pcgan_amd.parallel import GradSync
    _toy_step = globals()[{"dcgan": "_toy_step", "dcgan_pair": "_toy_pair_step"}.get(toy, "_toy_countergan_step")]

    def build():
        rec = _Recorder()
        netG, netD = _ToyNet(96, 1), _ToyNet(64, 2)       # identical start on every rank
        return rec, netG, netD, _ToySGD(netD, rec), _ToySGD(netG, rec)

    # per-rank shards of the global batch (integer-valued: every sum below is exact in fp32, so bit-equality is meaningful)
    shards = [(torch.full((4,), float(1 + r)), torch.full((4,), float(3 - r))) for r in range(world)]
    steps = 3

    # (a) eager data-parallel steps
    rec, eG, eD, oD, oG = build()
    dp = GradSync()
    for _ in range(steps):
        _toy_step(rec, eG, eD, oD, oG, *shards[rank], dp)
    dp.wait_all()

    # (b) the same steps as a captured segment program
    rec, gG, gD, oD2, oG2 = build()
    dp2 = GradSync()
    real, noise = shards[rank][0].clone(), shards[rank][1].clone()
    before = (gG.flat_params.clone(), gD.flat_params.clone())
    gs = GraphedStep(lambda d: _toy_step(rec, gG, gD, oD2, oG2, real, noise, d), {"real": real, "noise": noise}, [gG, gD], [oD2, oG2],
                     warmup=2, dp=dp2, capture=rec)
    ok = torch.equal(gG.flat_params, before[0]) and torch.equal(gD.flat_params, before[1])      # building it did not train
    ok = ok and float(oD2.steps) == 0 and float(gD.bn_running.abs().sum()) == 0
    # wait(G) | sync_now(D) | sync_then(G, Adam) cut the step into 4 segments, with the three exchanges between them
    # (the paired order starts with wait(G): its first segment holds no launch, the program keeps the same shape)
    ok = ok and len(gs.program) == 4 and [op is not None for _, op in gs.program] == [True, True, True, False]
    for _ in range(steps):
        gs.load(real=shards[rank][0], noise=shards[rank][1])
        gs.replay()
    dp2.wait_all()
    ok = ok and torch.equal(gG.flat_params, eG.flat_params) and torch.equal(gD.flat_params, eD.flat_params)
    ok = ok and torch.equal(gD.bn_running, eD.bn_running) and float(oG2.steps) == steps

    if toy not in ("dcgan", "dcgan_pair"):      # the single-process closed form below is the DCGAN toy's; (a) == (b) and (d) cover this order
        for t in (gG.flat_params, gD.flat_params):
            gathered = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            ok = ok and all(torch.equal(gathered[0], x) for x in gathered)
        q.put((rank, bool(ok)))
        dist.destroy_process_group()
        return
    # (c) one process, averaged gradients of all shards (what the replicas must equal)
    rec, sG, sD, oD3, oG3 = build()
    for _ in range(steps):
        sD.flat_grads.zero_()
        for r_, n_ in shards:
            sD.flat_grads.add_((r_.sum() * sD.flat_params + n_.sum() * sG.flat_params[:64]) / world)
        oD3.step()
        sG.flat_grads.zero_()
        for r_, n_ in shards:
            sG.flat_grads.add_((n_.sum() * sG.flat_params + sD.flat_params.sum()) / world)
        oG3.step()
    ok = ok and torch.allclose(gG.flat_params, sG.flat_params, rtol=1e-6) and torch.allclose(gD.flat_params, sD.flat_params, rtol=1e-6)

    # (d) replicas bit-identical
    for t in (gG.flat_params, gD.flat_params):
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        ok = ok and all(torch.equal(gathered[0], x) for x in gathered)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("toy", ["dcgan", "dcgan_pair", "countergan"])
def test_segment_program_world2_gloo(toy):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_segment_worker, args=(r, 2, port, q, toy)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)], res


def test_bench_launcher_starts_ranks_and_propagates_failure(tmp_path, monkeypatch):
    """`python bench.py --gpus N` outside torch.distributed.run: N child ranks with the torchrun environment, rank 0's stdout is
    the bench's stdout, a failing rank fails the run.  (A stand-in rank script: no GPU here.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\n"
                      "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
                      "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "print('{\"rank\": %d, \"world\": %d, \"argv\": \"%s\"}' % (r, w, ' '.join(sys.argv[1:])))\n"
                      "sys.exit(7 if (len(sys.argv) > 1 and sys.argv[1] == 'fail' and r == 1) else 0)\n")
    # the parent counts GPUs from the KFD topology (no HIP call); a runtime query there fails the test
    drv = ("import sys, torch; sys.path.insert(0, %r); import bench; bench.count_gpus_sysfs = lambda: %%d; "
           "torch.cuda.device_count = lambda: (_ for _ in ()).throw(AssertionError('the parent must not ask the HIP runtime')); "
           "bench.launch_ranks(3, sys.argv[1:], script=%r)" % (root, str(script)))
    ok = subprocess.run([sys.executable, "-c", drv % 8, "--steps", "2"], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stderr
    assert ok.stdout.strip() == '{"rank": 0, "world": 3, "argv": "--steps 2"}'          # only rank 0 reaches stdout
    assert '[rank 1] {"rank": 1' in ok.stderr and '[rank 2] {"rank": 2' in ok.stderr      # every rank's output carries its rank
    bad = subprocess.run([sys.executable, "-c", drv % 8, "fail"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 7 and "rank 1 exited with 7" in bad.stderr
    few = subprocess.run([sys.executable, "-c", drv % 1], capture_output=True, text=True, timeout=120)
    assert few.returncode != 0 and "exposes 1 GPU" in few.stderr
    assert "rank" not in few.stdout and '{"rank"' not in few.stderr                        # failed before any child started


def test_bench_launcher_rendezvous_timeout(tmp_path):
    """A rank that never arrives (stuck in the rendezvous, a hung collective) must not hang the run: after PCG_BENCH_TIMEOUT
    seconds the launcher stops the ranks it started and exits non-zero."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n"
                      "    sys.stderr.write('waiting for a peer that never comes\\n'); sys.stderr.flush(); time.sleep(600)\n")
    drv = ("import sys, torch; sys.path.insert(0, %r); import bench; bench.count_gpus_sysfs = lambda: 2; "
           "bench.launch_ranks(2, [], script=%r)" % (root, str(script)))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", drv], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, PCG_BENCH_TIMEOUT="3"))
    assert r.returncode == 124 and "still running after 3 s" in r.stderr and "[rank 1] waiting for a peer" in r.stderr
    assert time.time() - t0 < 60


def test_gpu_count_from_kfd_topology(tmp_path, monkeypatch):
    """count_gpus_sysfs reads /sys/class/kfd/kfd/topology/nodes/*/properties: nodes with simd_count > 0 are GPUs."""
    import bench
    n = bench.count_gpus_sysfs()
    assert n is None or n >= 0
    base = tmp_path / "nodes"
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):
        (base / str(i)).mkdir(parents=True)
        (base / str(i) / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    real_listdir, real_open = os.listdir, open
    monkeypatch.setattr(os, "listdir", lambda p: real_listdir(str(base)) if p == "/sys/class/kfd/kfd/topology/nodes" else real_listdir(p))
    import builtins
    monkeypatch.setattr(builtins, "open", lambda p, *a, **k: real_open(str(p).replace("/sys/class/kfd/kfd/topology/nodes", str(base)), *a, **k))
    assert bench.count_gpus_sysfs() == 3


def _syncbn_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pcgan_amd.parallel import GradSync
    dp = GradSync(sync_bn=True)
    # exact-BatchNorm exchange: each rank holds the fp64 sums (sum x, sum x^2 | sum dy, sum dy*xhat) of ITS shard; after the
    # all-reduce every rank holds the sums of the whole batch — what the statistics finalize then consumes (rows * world).
    # Integer-valued fp32 data: every partial sum is exact in fp64, so "2 ranks x B/2 == 1 rank x B" holds bit for bit.
    g = torch.Generator().manual_seed(5)
    B, C = 64, 16
    x = torch.randint(-50, 50, (B, 7, C), generator=g).float()
    shard = x[rank * (B // world):(rank + 1) * (B // world)].double().reshape(-1, C)
    sums = torch.cat([shard.sum(0), (shard * shard).sum(0)])
    dp.allreduce_sum_f64_(sums)
    full = x.double().reshape(-1, C)
    want = torch.cat([full.sum(0), (full * full).sum(0)])
    rows = full.shape[0]
    mean, var = sums[:C] / rows, sums[C:] / rows - (sums[:C] / rows) ** 2
    ok = torch.equal(sums, want) and torch.allclose(mean, full.mean(0), rtol=0, atol=1e-12) \
        and torch.allclose(var, full.var(0, unbiased=False), rtol=1e-12)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_exact_batchnorm_sums_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)], res
