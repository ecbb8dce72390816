"""GPU: scratch-buffer lifetime against captured HIP graphs.  A captured step has the addresses of its split-K slabs,
BatchNorm partial rows and weight-gradient slabs baked in; ops._scratch therefore keys scratch by (device, stream) and RETIRES an
outgrown buffer instead of freeing it.  Also: a DeviceRNG draw inside a capture must fail loudly (it would replay the same numbers)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def pcg():
    import pcgan_amd
    from pcgan_amd import dcgan  # noqa: F401
    return pcgan_amd


def _fresh(D, cfg, seed=5):
    torch.manual_seed(seed)
    netG, netD = D.Generator(cfg).to(DEV), D.Discriminator(cfg).to(DEV)
    netG.apply(D.weights_init); netD.apply(D.weights_init)
    return (netG, netD) + tuple(D.make_optimizers(netG, netD, cfg))


def test_larger_eager_op_after_capture_does_not_corrupt_replay(pcg):
    from pcgan_amd.nn import GraphedStep
    D, ops = pcg.dcgan, pcg.ops
    cfg = {"g_hidden": 32, "d_hidden": 32, "z_dim": 64}
    B = 32
    g = torch.Generator().manual_seed(3)
    reals = [(torch.rand(B, 1, 64, 64, generator=g) * 2 - 1).to(DEV) for _ in range(3)]
    noises = [torch.randn(B, 64, 1, 1, generator=g).to(DEV) for _ in range(3)]
    # eager trajectory
    netG, netD, crit, optD, optG = _fresh(D, cfg)
    for i in range(3):
        o = D.train_step(netG, netD, crit, optD, optG, reals[i], noises[i], cfg)
    want = ([o[k].item() for k in ("errD_real", "errD_fake", "errG")], netG.flat_params.clone(), netD.flat_params.clone())
    # captured, then a MUCH larger eager weight gradient (grows every scratch buffer of the main stream) and a burst of
    # allocations that would recycle any freed block, then replay
    netG, netD, crit, optD, optG = _fresh(D, cfg)
    s_real, s_noise = reals[0].clone(), noises[0].clone()
    gs = GraphedStep(lambda: D.train_step(netG, netD, crit, optD, optG, s_real, s_noise, cfg), {"real": s_real, "noise": s_noise},
                     [netG, netD], [optD, optG])
    retired_before = len(ops._ws_retired)
    small = ops.workspace(1 << 20, torch.device(DEV))
    geom = ops.conv_geom(256, 32, 32, 128, 256, 4, 4, 2, 1)
    x = torch.randn(256, 32, 32, 128, device=DEV)
    dy = torch.randn(256, geom.OH, geom.OW, 256, device=DEV)
    dw = torch.empty(256, 4, 4, 128, device=DEV)
    ops.conv2d_wgrad(geom, x, dy, dw, False)
    big = ops.workspace(1 << 20, torch.device(DEV))
    if big.data_ptr() != small.data_ptr():          # the main stream's buffer was outgrown: the old one must still be alive
        assert len(ops._ws_retired) > retired_before and any(t.data_ptr() == small.data_ptr() for t in ops._ws_retired)
    junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(16)]     # poison whatever the allocator hands out
    for i in range(3):
        gs.load(real=reals[i], noise=noises[i])
        o = gs.replay()
    torch.cuda.synchronize()
    del junk
    got = ([o[k].item() for k in ("errD_real", "errD_fake", "errG")], netG.flat_params, netD.flat_params)
    assert got[0] == want[0]
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])


def test_scratch_is_per_stream_and_retired_not_freed(pcg):
    ops = pcg.ops
    dev = torch.device(DEV)
    side = torch.cuda.Stream()
    a = ops.workspace(1 << 20, dev)
    with torch.cuda.stream(side):
        b = ops.workspace(1 << 20, dev)
        n0 = len(ops._ws_retired)
        c = ops.workspace(b.numel() + 1, dev)       # outgrow the side stream's buffer
        assert len(ops._ws_retired) == n0 + 1 and ops._ws_retired[-1] is b and c.numel() > b.numel()
    assert a.data_ptr() != b.data_ptr()             # side-stream work (GradSync.sync_then callbacks) never shares scratch with main
    assert ops.workspace(1 << 20, dev).data_ptr() == a.data_ptr()
    assert ops.workspace2(1 << 20, dev).data_ptr() != a.data_ptr()


def test_rng_draw_inside_capture_raises(pcg):
    ops = pcg.ops
    rng = ops.DeviceRNG(seed=3)
    rng.randn((16,), torch.device(DEV))             # fine outside a capture
    g = torch.cuda.CUDAGraph()
    with pytest.raises(pcg.PcgError, match="capture"):
        with torch.cuda.graph(g):
            torch.zeros(1, device=DEV).add_(1.0)
            rng.randn((16,), torch.device(DEV))


def test_cyclic_garbage_before_capture_does_not_abort(pcg):
    """The r02 process abort (gpurun_out/h17): Python's cyclic collector ran a finalizer that frees device memory while a stream
    was capturing.  Reference cycles holding device tensors, events and a finished graph are left uncollected right before the
    capture (collector thresholds lowered so that it WOULD run inside the capture if it were allowed to); GraphedStep must collect
    them first, keep the collector off while capturing and hand it back enabled."""
    import gc
    from pcgan_amd.nn import GraphedStep
    D = pcg.dcgan
    cfg = {"g_hidden": 16, "d_hidden": 16, "z_dim": 32}
    netG, netD, crit, optD, optG = _fresh(D, cfg)
    real = (torch.rand(8, 1, 64, 64) * 2 - 1).to(DEV)
    noise = torch.randn(8, 32, 1, 1).to(DEV)

    class Node:
        pass
    old = gc.get_threshold()
    gc.collect()
    try:
        gc.set_threshold(5, 1, 1)
        gc.disable()
        for _ in range(64):                         # cycles that own device memory, events and a captured graph
            a, b = Node(), Node()
            a.peer, b.peer = b, a
            a.t = torch.empty(1 << 16, device=DEV)
            b.e = torch.cuda.Event(enable_timing=True)
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0):
            torch.zeros(4, device=DEV).add_(1.0)
        c = Node(); c.me = c; c.g = g0
        del a, b, c, g0
        gc.enable()
        assert gc.isenabled()
        gs = GraphedStep(lambda: D.train_step(netG, netD, crit, optD, optG, real, noise, cfg), {"real": real, "noise": noise},
                         [netG, netD], [optD, optG])
        assert gc.isenabled()                       # handed back
        o = gs.replay()
        torch.cuda.synchronize()
        assert all(torch.isfinite(o[k]).item() for k in ("errD_real", "errD_fake", "errG"))
    finally:
        gc.set_threshold(*old)
        gc.enable()


def test_collector_is_handed_back_when_capture_cannot_begin(pcg, monkeypatch):
    """_HipGraphCapture.begin disables the collector; if torch.cuda.graph(...).__enter__ raises, end() never runs — begin itself
    must re-enable it (r02 residual)."""
    import gc
    from pcgan_amd import nn as N
    cap = N._HipGraphCapture(torch.device(DEV))

    class Boom:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            raise RuntimeError("capture refused")
    monkeypatch.setattr(torch.cuda, "graph", Boom)
    assert gc.isenabled()
    with pytest.raises(RuntimeError, match="capture refused"):
        cap.begin()
    assert gc.isenabled()


def test_native_rccl_exports_one_rank(pcg):
    """pcg_dp_* (RCCL behind the C ABI) on a one-rank communicator: in-stream average, side-stream begin / record / wait with
    follow-up work queued behind the reduction, float64 sum, broadcast — and the ordering against the producer stream."""
    import ctypes
    import os
    import torch.distributed as dist
    from pcgan_amd import parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29549")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        dp = parallel.GradSync(always_exchange=True)
        assert dp.native and dp.rccl_ranks() == 1
        lib = pcg.load()
        assert lib.pcg_dp_world() == 1 and lib.pcg_dp_rank() == 0 and lib.pcg_dp_side_stream()

        class Net:
            pass
        net = Net()
        net.flat_grads = torch.arange(1 << 20, dtype=torch.float32, device=DEV)
        want = net.flat_grads.clone()
        dp.sync_now(net)                                     # mean over one rank: unchanged
        assert torch.equal(net.flat_grads, want)
        # side stream: the reduction must see the producer's writes, the follow-up must see the reduction, the consumer both
        net.flat_grads.mul_(2.0)                             # producer work on the main stream
        out = torch.zeros_like(net.flat_grads)
        dp.sync_then(net, lambda: out.copy_(net.flat_grads).add_(1.0))
        dp.wait(net)
        assert torch.equal(out, want * 2 + 1)
        t = torch.full((256,), 0.1, dtype=torch.float64, device=DEV)
        dp.allreduce_sum_f64_(t)
        assert torch.equal(t, torch.full((256,), 0.1, dtype=torch.float64, device=DEV))
        b = torch.randn(1000, device=DEV)
        b0 = b.clone()
        dp.broadcast_(b, 0)
        assert torch.equal(b, b0)
        torch.cuda.synchronize()
    finally:
        parallel.shutdown()
        assert pcg.load().pcg_dp_world() == 0
        dist.destroy_process_group()


def test_exact_batchnorm_mode_one_rank_equals_per_replica(pcg):
    """pcg_dp_sync_batchnorm on a one-rank communicator: every BatchNorm-family call takes the three-step path (this rank's sums
    in the usual fixed order -> fp64 all-reduce -> finalize of ONE row of global sums with rows*world rows).  With world = 1 the
    all-reduce is the identity and the sums are added in the same order, so a full DCGAN step — conv-epilogue statistics,
    stand-alone statistics, both BatchNorm-backward forms — is BIT-identical to the default mode."""
    import os
    import torch.distributed as dist
    from pcgan_amd import parallel
    from pcgan_amd.nn import GraphedStep
    D = pcg.dcgan
    cfg = {"g_hidden": 32, "d_hidden": 32, "z_dim": 64}
    g = torch.Generator().manual_seed(8)
    real = (torch.rand(24, 1, 64, 64, generator=g) * 2 - 1).to(DEV)
    noise = torch.randn(24, 64, 1, 1, generator=g).to(DEV)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29551")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    res = {}
    try:
        for sync in (False, True):
            dp = parallel.GradSync(always_exchange=True, sync_bn=sync)
            netG, netD, crit, optD, optG = _fresh(D, cfg)
            for _ in range(2):
                # pair=False: the exact mode has no grouped form (the step falls back to two passes), so the comparison runs both that way
                o = D.train_step(netG, netD, crit, optD, optG, real, noise, cfg, dp=dp, skip_dead_d_wgrad=False, pair=False)
            dp.wait_all()
            res[sync] = ([o[k].item() for k in ("errD_real", "errD_fake", "errG")], netG.flat_params.clone(), netD.flat_params.clone(),
                         [b.clone() for b in netD.buffers()] + [b.clone() for b in netG.buffers()])
            if sync:
                with pytest.raises(pcg.PcgError, match="eagerly"):
                    GraphedStep(lambda d: None, {"real": real}, [netG, netD], [optD, optG], dp=dp)
        assert res[True][0] == res[False][0]
        assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])
        assert all(torch.equal(a, b) for a, b in zip(res[True][3], res[False][3]))
    finally:
        parallel.shutdown()
        dist.destroy_process_group()
